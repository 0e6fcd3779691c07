"""The driver's contract with bench.py: ONE JSON line on stdout with the agreed keys (a short two-layer debug run; the
numbers of such a run are marked INVALID by bench.py itself and are not looked at here)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--layers", "2",
                        "--no-decode-leg", "--cpu-layers", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 1 and r["warmup"] == 0 and r["higher_is_better"] is True
    assert r["metric"] == "latent_frames_per_sec" and r["unit"] == "frames/s" and r["value"] > 0 and r["vs_baseline"] is None
    assert r["dtype"] == "bf16" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert r["config"]["INVALID"]                                  # two layers: bench.py says so itself
    rf = r["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf and rf["launches"] > 0
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"] and cb["unit"]


def test_bench_two_ranks_report_the_ranks_an_all_gather_saw_and_the_exchange_used():
    """`bench.py --gpus 2` as the driver launches it, both ranks on the one GPU of the test box (IFX_BENCH_SHARE_GPU=1: gloo instead
    of RCCL, everything else the N > 1 path): the line carries `rccl_ranks` counted from a collective and the per-layer K/V exchange
    that ran — the peer-store path after its self-test between the two processes."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, IFX_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--layers", "2", "--no-decode-leg", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["value"] > 0
    assert r["rccl_ranks"] == 2 and r["sp_exchange"] == "peer_store", (r["rccl_ranks"], r["sp_exchange"])


def test_bench_magi_leg_runs_the_schedule_through_the_model():
    """`bench.py --magi-leg fp8` (the config 5 leg alone) on a cut-down stack: four layers, the first step of every schedule stage —
    seven forwards of 1 .. 4 chunks, three of them with the clean chunk in front — through HipVideoDiTModel under the emulated cp = 8
    rank; the JSON carries the counts the schedule implies and says INVALID."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--magi-leg", "fp8", "--magi-steps", "1", "--layers", "4"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["denoise_forwards"] == 7 and d["chunk_forwards"] >= 1 + 2 + 3 + 4 + 4 + 3 + 2
    assert "INVALID" in d and "partial schedule" in d["INVALID"] and d["fp8_quant"] is True
    assert d["roofline"]["launches"] > 0 and d["gemm_fp8_ms"] > 0 and d["ms_clip_rank"] > 0
