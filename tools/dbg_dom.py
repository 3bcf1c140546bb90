import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctts_amd import kernels as K
from ctts_amd.synthetic import CANONICAL_SRC_LENS
dev="cuda"; B,T=16,1024; M=B*T
def t(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters*1e3
lens = torch.tensor([8*v for v in CANONICAL_SRC_LENS], dtype=torch.int32, device=dev)
tmap = K.row_tile_map(lens, T, 0, M)
x = torch.randn(B,T,256,device=dev); wf = torch.randn(1024,2304,device=dev)*0.02; C = torch.empty(B,T,1024,device=dev); Z = torch.empty_like(C)
seed = torch.zeros(1,dtype=torch.int64,device=dev)
def mk(bias, sk): return lambda: K.gemm(x,wf,C,M,1024,2304,256,2304,1024,True,True,conv=(T,4,256),alpha=9**-0.5,bias=bias,Z=Z,ldz=1024,act=K.ACT_GELU,p_drop=0.1,seed=seed,drop_offset=1,row_lens=lens,row_T=T,row_halo=0,tile_map=tmap,use_sk=sk)
b0 = torch.zeros(1024,device=dev); b1 = torch.randn(1024,device=dev)*0.1
for name, bias in (("bias=0", b0), ("bias=rand", b1), ("bias=0 again", b0)):
    for sk in (False, True):
        print(name, "sk" if sk else "old", f"{t(mk(bias, sk)):.1f} us", f"{t(mk(bias, sk), iters=100):.1f} us (100 it)", f"{t(mk(bias, sk), iters=300):.1f} us (300 it)")
