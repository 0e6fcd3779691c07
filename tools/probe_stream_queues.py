"""Which HIP streams can make progress while a kernel SPINS on another one?  (HIP streams share a small number of hardware queues;
a spinning wait kernel at the head of a hardware queue blocks every stream mapped onto the same queue — the hazard of device-side
waits on more than a few streams: the peer-store exchange of a sequence-parallel rank waits on device, and the paired forwards run
two launch chains + two exchange streams.)
For torch's stream pool: spin on stream i (ifx_peer_wait on a pinned host flag, bounded), launch a trivial kernel on stream j, see
whether it completes while i is still spinning.  usage: tools/probe_stream_queues.py [n_streams=8] [GPU_MAX_HW_QUEUES to set from inside python]"""
import os, sys, time
if len(sys.argv) > 2:                       # probe_stream_queues.py n Q: set GPU_MAX_HW_QUEUES=Q from INSIDE python, before torch is imported
    os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[2]
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inferix_amd import hip_ops as ops

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(unset)"))
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(n)]
    names = ["current"] + [f"pool{i}" for i in range(n)]
    flag = torch.zeros(16, dtype=torch.int32).pin_memory()
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    x = torch.zeros(1024, device=dev)
    torch.cuda.synchronize()
    blocked = {}
    for i, si in enumerate(streams):
        flag.zero_()
        with torch.cuda.stream(si):
            ops.peer_wait(flag.data_ptr(), 1, 1, 3000, status)           # spins until the host sets flag[0] (or 3 s)
        time.sleep(0.02)
        evs = []
        for j, sj in enumerate(streams):
            if j == i:
                evs.append(None)
                continue
            with torch.cuda.stream(sj):
                x.add_(1.0)
                e = torch.cuda.Event()
                e.record()
            evs.append(e)
        time.sleep(0.1)
        blocked[i] = [j for j, e in enumerate(evs) if e is not None and not e.query()]
        flag[0] = 1                                                       # release
        torch.cuda.synchronize()
        assert int(status.item()) == 0, "the wait timed out"
        print(f"spin on {names[i]:8s}: blocked = {[names[j] for j in blocked[i]]}", flush=True)
    groups = []
    for i in range(len(streams)):
        for g in groups:
            if g[0] in blocked[i] or i in blocked[g[0]]:
                g.append(i)
                break
        else:
            groups.append([i])
    print("hardware-queue groups:", [[names[i] for i in g] for g in groups])

main()
