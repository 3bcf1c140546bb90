"""CPU: host-side logic added in round 6 that needs no GPU - the C-ABI collective's constants against rccl.h and its error path
without RCCL devices, backend selection of the multi-rank test workers, the weight-epoch stamp of the step caches, the stream tags of
the deferred-sum sink."""
import ctypes
import os
import re
import sys

import pytest
import torch

import ctts_amd  # noqa: F401
from ctts_amd import _lib
from ctts_amd import kernels as K
from ctts_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_comm_constants_match_the_rccl_header_this_image_ships():
    """csrc/comm.hip binds RCCL by dlopen and therefore restates three constants of rccl.h (ncclFloat32 = 7, ncclAvg = 4,
    NCCL_UNIQUE_ID_BYTES = 128) and four prototypes: pin them to the header (skipped where ROCm's headers are absent)."""
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("rccl.h not installed")
    txt = open(hdr).read()
    assert re.search(r"ncclFloat32\s*=\s*7\b", txt) and re.search(r"ncclAvg\s*=\s*4\b", txt)
    assert re.search(r"#define\s+NCCL_UNIQUE_ID_BYTES\s+128\b", txt) and _lib.COMM_ID_BYTES == 128
    assert re.search(r"ncclCommInitRank\(ncclComm_t\*\s*comm,\s*int\s+nranks,\s*ncclUniqueId\s+commId,\s*int\s+rank\)", txt)
    assert re.search(r"ncclAllReduce\(const void\*\s*sendbuff,\s*void\*\s*recvbuff,\s*size_t\s+count,\s*ncclDataType_t\s+datatype,\s*"
                     r"ncclRedOp_t\s+op,\s*ncclComm_t\s+comm,\s*hipStream_t\s+stream\)", txt)
    src = open(os.path.join(ROOT, "comprehensive-transformer-tts_amd", "csrc", "comm.hip")).read()
    assert "kFloat32 = 7" in src and "kAvg = 4" in src
    h = open(os.path.join(ROOT, "include", "ctts.h")).read()
    assert "#define CTTS_COMM_ID_BYTES 128" in h


def test_comm_entry_points_reject_bad_arguments_without_touching_rccl():
    lib = _lib.load()
    comm = ctypes.c_void_p()
    uid = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
    assert lib.ctts_comm_create(ctypes.byref(comm), 2, 5, uid) != 0            # rank >= nranks
    assert b"ctts_comm_create" in lib.ctts_last_error()
    assert lib.ctts_comm_create(None, 1, 0, uid) != 0
    assert lib.ctts_comm_unique_id(None) != 0
    assert lib.ctts_allreduce_mean(None, 0, None, None) == 0                     # nothing to reduce
    assert lib.ctts_allreduce_mean(None, 16, None, None) != 0
    assert lib.ctts_comm_destroy(None) == 0


def test_dp_workers_pick_rccl_only_when_every_rank_gets_its_own_gpu(monkeypatch):
    """tests/dp_worker.py / ddp_worker.py take backend "nccl" + cuda:{LOCAL_RANK} when the box has >= WORLD_SIZE GPUs
    (the 2-rank = 1-rank equality test then runs over RCCL by itself) and today's gloo + cuda:0 otherwise."""
    import dp_worker
    monkeypatch.delenv("CTTS_TEST_BACKEND", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert dp_worker.pick_backend(1, 2) == ("gloo", torch.device("cuda:0"))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert dp_worker.pick_backend(1, 2) == ("nccl", torch.device("cuda:1"))
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert dp_worker.pick_backend(11, 16 // 2) == ("nccl", torch.device("cuda:3"))
    assert dp_worker.pick_backend(0, 1) == ("gloo", torch.device("cuda:0"))          # a single rank needs no collective backend
    monkeypatch.setenv("CTTS_TEST_BACKEND", "gloo")
    assert dp_worker.pick_backend(1, 2) == ("gloo", torch.device("cuda:0"))
    monkeypatch.setenv("CTTS_TEST_BACKEND", "nccl")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(RuntimeError, match="needs 2 GPUs"):
        dp_worker.pick_backend(1, 2)


def test_weight_caches_are_stamped_with_the_raw_update_epoch():
    """dp.FlatAdam updates parameters through raw pointers - `w._version` never moves - so the step caches carry
    kernels.WEIGHTS_EPOCH as well: an entry made before a FlatAdam.step() is not handed out after it."""
    w = torch.nn.Parameter(torch.zeros(4, 3, 5))
    ops._DGRAD_W.clear()
    ops._DGRAD_W[w.data_ptr()] = (torch.zeros(3, 20), ops._wstamp(w))
    assert ops._DGRAD_W.take(w, (3, 20)) is not None
    ops._DGRAD_W[w.data_ptr()] = (torch.zeros(3, 20), ops._wstamp(w))
    K.WEIGHTS_EPOCH[0] += 1                                   # what dp.FlatAdam.step does
    assert ops._DGRAD_W.take(w, (3, 20)) is None
    ops._DGRAD_W[w.data_ptr()] = (torch.zeros(3, 20), ops._wstamp(w))
    with torch.no_grad():
        w.add_(1.0)                                           # an in-place update through torch: the autograd version moves
    assert ops._DGRAD_W.take(w, (3, 20)) is None
    # announcements of tensors that are gone do not survive the next preparation
    ops._PLANES["want_fwd"].add(12345)
    ops.prepare_dgrad_weights([])
    assert 12345 not in ops._PLANES["want_fwd"]


def test_partial_sink_remembers_producer_streams_only_for_device_partials():
    """a flush must be ordered behind every stream that produced pending partials.  On the CPU there is nothing
    to order - the bookkeeping must stay empty and a flush of nothing must not touch a stream."""
    sink = K.PartialSink()
    assert sink.streams == [] and sink.tasks == []
    sink.flush()                                              # nothing pending: no launch, no stream query
    src, dst = torch.zeros(2, 8), torch.zeros(8)
    sink.tasks.append((src.data_ptr(), dst.data_ptr(), 8, 8, 2, 1.0))      # as add() would, minus the launch at flush
    sink.keep.append((src, dst))
    assert sink.streams == []


def test_attached_planes_are_validated_against_the_tensor_they_were_made_for():
    """kernels.attach_planes / planes_of (the producers' side channel): the set is handed out only for the SAME storage address, element
    count and autograd version; it survives the trip through autograd.Function boundaries in both directions (the property the whole
    scheme rests on: autograd keeps a tensor's Python object)."""
    t = torch.zeros(4, 64)
    pl = torch.zeros(4, 2, 3, 32, dtype=torch.bfloat16)
    assert K.planes_of(t) is None
    K.attach_planes(t, pl)
    assert K.planes_of(t) is pl
    v = t.view(2, 2, 64)                                # another Python object over the same memory: nothing attached
    assert K.planes_of(v) is None
    K.attach_planes(v, pl)                              # what _BatchNormAct does for the reshaped tensor it returns
    assert K.planes_of(v) is pl
    t.add_(1.0)                                         # written since: void (the view shares the version counter)
    assert K.planes_of(t) is None and K.planes_of(v) is None
    t2 = torch.zeros(4, 64)
    t2._ctts_planes = (torch.zeros(3, dtype=torch.bfloat16), t2.data_ptr(), t2._version, t2.numel())      # wrong size
    assert K.planes_of(t2) is None

    class Producer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            y = x * 2
            K.attach_planes(y, torch.zeros(y.numel() * 3, dtype=torch.bfloat16))
            return y

        @staticmethod
        def backward(ctx, g):
            seen.append(("producer.backward", K.planes_of(g) is not None))
            return g * 2

    class Consumer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, y):
            seen.append(("consumer.forward", K.planes_of(y) is not None))
            return y + 1

        @staticmethod
        def backward(ctx, g):
            out = g.clone()
            K.attach_planes(out, torch.zeros(out.numel() * 3, dtype=torch.bfloat16))
            return out

    seen = []
    x = torch.randn(2, 32, requires_grad=True)
    Consumer.apply(Producer.apply(x)).sum().backward()
    assert seen == [("consumer.forward", True), ("producer.backward", True)], seen


def test_consumer_takes_planes_needs_a_device_tensor_and_the_plane_kernel_shape():
    x = torch.zeros(16, 1024, 256)
    assert not ops.consumer_takes_planes(x, 1024, 9)            # host tensor: the product has no CPU path
    prev = K.PRODUCER_PLANES
    try:
        K.PRODUCER_PLANES = False
        assert not ops.consumer_takes_planes(x, 1024, 9)
    finally:
        K.PRODUCER_PLANES = prev


def test_a_plain_c_host_binds_the_abi(tmp_path):
    """include/ctts.h is a C header (`gcc -std=c99`) and libctts_hip.so a plain C-ABI library: a C program with no Python and no torch in
    it links the library and calls it (tests/native/c_host_abi.c: version, the collective's argument checks, the error text)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "c_host_abi")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "c_host_abi.c"),
                    "-o", exe, "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                   check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c host ok" in r.stdout, (r.stdout, r.stderr)
