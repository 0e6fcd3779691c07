import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0)
for fp8 in (False, True):
    r = bench.magi_cp8_emulated_leg(dev, fp8_quant=fp8)
    print(json.dumps({k: r[k] for k in ("fp8_quant", "ms_per_denoise_forward_rank", "attn_ms", "gemm_ms", "gemm_tflops", "gemm_fp8_ms", "gemm_fp8_tflops")}))
