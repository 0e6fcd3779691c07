#!/usr/bin/env python3
"""Does RCCL accept two ranks on ONE device (so that the N>1 RCCL path can be exercised on a 1-GPU box)?  Tries, in separate
2-process groups: (a) plain nccl init with both ranks on cuda:0; then all_gather_into_tensor + all_to_all_single.  Prints one JSON line.
usage: python tools/probe_rccl_one_gpu.py"""
import json
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, env, ret):
    os.environ.update(env)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    try:
        import datetime
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=60))
        x = torch.full((4,), float(rank + 1), device="cuda")
        out = torch.empty(4 * world, device="cuda")
        dist.all_gather_into_tensor(out, x)
        torch.cuda.synchronize()
        ret[rank] = ("ok", out.tolist())
        dist.destroy_process_group()
    except Exception as exc:  # noqa: BLE001
        ret[rank] = ("error", str(exc)[:300])


def main():
    res = {}
    for name, env in (("plain", {}), ("ignore_dup", {"NCCL_IGNORE_DUPLICATE_GPU": "1", "RCCL_IGNORE_DUPLICATE_GPU": "1"})):
        ret = mp.Manager().dict()
        try:
            mp.spawn(worker, args=(2, _free_port(), env, ret), nprocs=2, join=True)
        except Exception as exc:  # noqa: BLE001
            ret["spawn"] = ("error", str(exc)[:300])
        res[name] = dict(ret)
    # world_size 1 over RCCL (always possible): the collective calls and stream ordering under the real backend
    ret = mp.Manager().dict()
    mp.spawn(worker, args=(1, _free_port(), {}, ret), nprocs=1, join=True)
    res["world1"] = dict(ret)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
