"""inferix/core/memory/utils.py:12-13,74-84: the two helpers the example scripts call.  On a 288 GB MI355X the low-memory
branches they guard (< 40 GB free) never trigger."""
import torch


def gpu():
    return torch.device(f"cuda:{torch.cuda.current_device()}")


def get_cuda_free_memory_gb(device=None):
    device = gpu() if device is None else device
    stats = torch.cuda.memory_stats(device)
    free, _ = torch.cuda.mem_get_info(device)
    inactive = stats.get("reserved_bytes.all.current", 0) - stats.get("active_bytes.all.current", 0)
    return (free + inactive) / (1024 ** 3)
