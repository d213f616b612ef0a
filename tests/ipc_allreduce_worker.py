"""Worker of tests/test_gpu_ipc_allreduce.py: one of several PROCESSES that share GPU 0 (launched by torch.distributed.run).
Sets up the mailbox all-reduce across the processes (handles all-gathered over gloo) and checks a few hundred reductions
against numpy on the gathered contributions."""
import os
import sys

os.environ.pop("NCCL_DEBUG", None)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
import jutul_amd as ja

tmo = float(os.environ.get("JH_TEST_COMM_TIMEOUT_S", "600"))
ctx = ja.HIPContext(0, comm_timeout_ms=int(tmo * 1000))  # every process on the same device; the option bounds the in-kernel waits
ctx.comm_init_ipc_only(world, rank)
handles = [None] * world
dist.all_gather_object(handles, ctx.comm_ipc_export())
ok = ctx.comm_ipc_attach(handles)
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
assert int(flag[0]) == 1, "mailbox self-test failed on some rank"
ctx.comm_ipc_enable(True)
# all ranks of this test share GPU 0 and none is CU-masked: the library has seen that (PCI bus ids exchanged by the attach) and
# refuses the promise "every rank has compute units of its own" instead of letting chip-filling kernels wait for each other
if os.environ.get("JH_TEST_EXCLUSIVE_REFUSAL") == "1":   # (tests/test_gpu_zz_round6.py: written in a round without a GPU)
    assert ctx.comm_devices_distinct() is False
    try:
        ctx.comm_set_exclusive(True)
        raise AssertionError("comm_set_exclusive(True) was accepted for ranks sharing an unmasked device")
    except ja.JutulHIPError as e:
        assert "same device" in str(e) and "CU mask" in str(e), str(e)
    assert not ctx.comm_info()["consumer_allreduce"]
    print("EXCLUSIVE_REFUSED_OK", world, flush=True)
rng = np.random.default_rng(100 + rank)
for t in range(300):
    n = 1 + t % 8
    op = "max" if t % 3 == 0 else "sum"
    mine = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4)
    allv = [None] * world
    dist.all_gather_object(allv, mine)
    got = ctx.allreduce(mine, op)
    acc = allv[0].copy()
    for r in range(1, world):   # rank order, like the kernel
        acc = np.maximum(acc, allv[r]) if op == "max" else acc + allv[r]
    assert np.array_equal(got, acc), (t, op, got, acc)
# every rank must hold identical bits
res = [None] * world
dist.all_gather_object(res, got.tobytes())
assert all(r == res[0] for r in res)
dist.barrier()
if os.environ.get("JH_TEST_TIMEOUT") == "1":
    # the last rank never enters the next reduction: every other rank must get an error naming it instead of hanging
    import time
    if rank == world - 1:
        time.sleep(tmo + 3.0)
    else:
        t0 = time.time()
        try:
            ctx.allreduce(np.ones(2), "sum")
            raise AssertionError("the all-reduce returned although a rank was missing")
        except ja.JutulHIPError as e:
            msg = str(e)
            assert "timed out" in msg and f"waiting for rank {world - 1}" in msg and f"rank {rank} of {world}" in msg, msg
        assert time.time() - t0 < tmo + 2.5
        assert ctx.comm_info()["timeouts"] >= 1
    dist.barrier()
    if rank == 0:
        print("IPC_TIMEOUT_OK", world, flush=True)
    os._exit(0)  # the communicator is unusable after a time-out: no orderly teardown
ctx.comm_finalize()
dist.barrier()
if rank == 0:
    print("IPC_ALLREDUCE_OK", world, flush=True)
