"""debug: does a forked side stream keep its dependency under hipGraph capture on this stack?"""
import torch
dev = "cuda"
cap, side = torch.cuda.Stream(), torch.cuda.Stream()
inp = torch.zeros(4096, 4096, device=dev)
w = torch.randn(4096, 4096, device=dev) * 0.01
out = {}

def body(mode):
    x = inp
    for _ in range(6):
        x = torch.tanh(x @ w + 1.0)               # slow producer on the capture stream
    cur = torch.cuda.current_stream()
    if mode == "fork":
        side.wait_stream(cur)
    elif mode == "event":
        ev = torch.cuda.Event(); ev.record()
    filler = x
    for _ in range(6):
        filler = torch.tanh(filler @ w)           # more work on the main stream after the fork point
    if mode == "event":
        side.wait_event(ev)
    with torch.cuda.stream(side):
        y = (x * 3.0).sum(1)                      # consumer on the side stream
    cur.wait_stream(side)
    return y + 0.0, filler

for mode in ("fork", "event"):
    torch.cuda.synchronize()
    with torch.cuda.stream(cap):
        for _ in range(2):
            body(mode)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
        y, f = body(mode)
    res = []
    for k in range(3):
        inp.fill_(0.1 * (k + 1))
        g.replay()
        torch.cuda.synchronize()
        ref_x = inp
        for _ in range(6):
            ref_x = torch.tanh(ref_x @ w + 1.0)
        ref = (ref_x * 3.0).sum(1)
        res.append(float((y - ref).abs().max()))
    print(mode, "max |replayed - eager| per replay:", res)
