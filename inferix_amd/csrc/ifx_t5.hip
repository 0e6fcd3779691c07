// umT5 text-encoder kernels, gfx950 only: the two pieces of a T5 block that are not a plain linear or an
// RMS norm — bidirectional self-attention with the additive relative-position bias and key-padding mask, and the gated GELU.
//
// Replaces `T5Attention.forward` between its q/k/v and o linears (inferix/models/wan_base/text_encoder/t5.py:96-117) together
// with `T5RelativeEmbedding.forward` (t5.py:235-245: the [1, heads, L, L] bias tensor is never built, the kernel indexes a
// per-head table of 2L-1 relative offsets), and `fc1(x) * GELU(gate(x))` of `T5FeedForward.forward` (t5.py:138-139, 50-52).
//
// Attention design: L <= 512 keys of 64 channels — the whole K (64 KiB) and V^T (64 KiB) of one head fit a CU's LDS, so a
// workgroup (4 waves x 32 queries) stages them once and every wave runs two passes over the keys with MFMA 32x32x16:
//   pass 1  S^T = K Q^T per 32-key block (keys in registers, queries in lanes) -> row max / row sum in fp32;
//   pass 2  S^T again, p = bf16(exp(s - m) / l) exactly as `softmax(attn.float()).type_as(attn)` rounds it, and the P
//           operand of the PV product is fed straight from the accumulators: the contraction slot (hi, j) of an MFMA k-step
//           is relabelled to the key the accumulator register holds, and V^T is read from LDS in that same order.
// Rounding points as upstream: s = bf16(bf16(q.k) + bias) (no 1/sqrt(d) scaling), masked keys = finfo(bf16).min.
#include "ifx_common.h"

namespace ifx {
namespace t5 {

constexpr int HD = 64;

struct AttnArgs {
  const unsigned short* q;
  const unsigned short* k;
  const unsigned short* v;
  unsigned short* out;
  const unsigned short* bias;      // [heads][2L-1]: bias of relative offset (key - query) + L - 1
  const int* seq_lens;             // [batch] valid keys per prompt (device)
  int ldq, ldk, ldv, ldo;          // row strides (elements)
  int L, heads, q_tiles;
};

__global__ __launch_bounds__(256) void t5_attention_kernel(AttnArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int L = A.L;
  const int VPB = L * 2 + 8;                                  // V^T row pitch in bytes (bank spread for 8-byte reads)
  unsigned char* Ks = smem;                                   // [L][64] bf16, 16-byte chunks XOR-swizzled by key & 7
  unsigned char* Vt = smem + L * 128;                         // [64][L] bf16
  unsigned short* bt = reinterpret_cast<unsigned short*>(Vt + HD * VPB);   // [2L-1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  int rem = blockIdx.x;
  const int qt = rem % A.q_tiles;
  rem /= A.q_tiles;
  const int h = rem % A.heads, b = rem / A.heads;
  const int seq = A.seq_lens[b];
  const size_t row0 = (size_t)b * L;

  for (int idx = tid; idx < L * 8; idx += 256) {
    const int key = idx >> 3, c = idx & 7;
    const u16x8 kv = *reinterpret_cast<const u16x8*>(A.k + (row0 + key) * A.ldk + h * HD + c * 8);
    *reinterpret_cast<u16x8*>(Ks + key * 128 + ((c ^ (key & 7)) << 4)) = kv;
    const u16x8 vv = *reinterpret_cast<const u16x8*>(A.v + (row0 + key) * A.ldv + h * HD + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) *reinterpret_cast<unsigned short*>(Vt + (c * 8 + e) * VPB + key * 2) = vv[e];
  }
  for (int i = tid; i < 2 * L - 1; i += 256) bt[i] = A.bias[(size_t)h * (2 * L - 1) + i];
  __syncthreads();

  const int q = qt * 128 + wave * 32 + l31;
  const int qc = min(q, L - 1);
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(A.q + (row0 + qc) * A.ldq + h * HD + ks * 16 + hi * 8);

  const float kMin = -3.3895313892515355e38f;                 // torch.finfo(torch.bfloat16).min
  const int nkb = L / 32;
  auto scores = [&](int kb, float (&s)[16]) {                 // s[r]: key kb*32 + (r/4)*8 + hi*4 + r%4, query = this lane's
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int key = kb * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ks + key * 128 + (((ks * 2 + hi) ^ (key & 7)) << 4));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = kb * 32 + (r >> 2) * 8 + hi * 4 + (r & 3);
      const float bias = kk < seq ? bf2f(bt[kk - qc + L - 1]) : kMin;
      s[r] = rbf(rbf(acc[r]) + bias);
    }
  };

  // ---- pass 1: row maximum and row sum (this lane sees half of the keys of its query; lane ^ 32 the other half)
  float m = kMin, l = 0.f;
  for (int kb = 0; kb < nkb; ++kb) {
    float s[16];
    scores(kb, s);
    float bm = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) bm = fmaxf(bm, s[r]);
    const float mn = fmaxf(m, bm);
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += __expf(s[r] - mn);
    l = l * __expf(m - mn) + acc;
    m = mn;
  }
  {
    const float mo = __shfl_xor(m, 32, 64), lo = __shfl_xor(l, 32, 64);
    const float mt = fmaxf(m, mo);
    l = l * __expf(m - mt) + lo * __expf(mo - mt);
    m = mt;
  }
  const float inv_l = 1.f / l;

  // ---- pass 2: P V with P from the accumulators
  f32x16 o[2];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[hb][r] = 0.f;
  for (int kb = 0; kb < nkb; ++kb) {
    float s[16];
    scores(kb, s);
    unsigned short p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = f2bf(__expf(s[r] - m) * inv_l);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      u16x8 pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = p[(2 * s2 + (j >> 2)) * 4 + (j & 3)];
      const int k0 = kb * 32 + (2 * s2) * 8 + hi * 4;         // slots 0..3 -> keys k0.., slots 4..7 -> keys k0 + 8..
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const unsigned char* vr = Vt + (hb * 32 + l31) * VPB;
        const u16x4 a0 = *reinterpret_cast<const u16x4*>(vr + k0 * 2);
        const u16x4 a1 = *reinterpret_cast<const u16x4*>(vr + (k0 + 8) * 2);
        const u16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        o[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, pb), o[hb],
                                                        0, 0, 0);
      }
    }
  }
  if (q >= L) return;
  unsigned short* op = A.out + (row0 + q) * A.ldo + h * HD;
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u16x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = f2bf(o[hb][4 * g + e]);
      *reinterpret_cast<u16x4*>(op + hb * 32 + g * 8 + hi * 4) = ov;
    }
}

// h = fc1 * GELU(gate), GELU = 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) evaluated op by op in bf16 like the
// reference's module (t5.py:50-52); `gf` holds [rows][2*ffn] = (gate | fc1) from one fused GEMM.
__global__ __launch_bounds__(256) void gated_gelu_kernel(const unsigned short* __restrict__ gf, unsigned short* __restrict__ h,
                                                         long long total8, int ffn8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total8) return;
  const long long row = i / ffn8;
  const int c = (int)(i - row * ffn8);
  const u16x8 gv = *reinterpret_cast<const u16x8*>(gf + (row * 2 * ffn8 + c) * 8);
  const u16x8 fv = *reinterpret_cast<const u16x8*>(gf + (row * 2 * ffn8 + ffn8 + c) * 8);
  u16x8 ov;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = bf2f(gv[e]);
    const float x3 = rbf(x * x * x);
    const float t1 = rbf(0.044715f * x3);
    const float t2 = rbf(x + t1);
    const float t3 = rbf(0.7978845608028654f * t2);
    const float t4 = rbf(tanhf(t3));
    const float t5 = rbf(1.0f + t4);
    const float t6 = rbf(0.5f * x);
    const float g = rbf(t6 * t5);
    ov[e] = f2bf(bf2f(fv[e]) * g);
  }
  *reinterpret_cast<u16x8*>(h + i * 8) = ov;
}

}  // namespace t5
}  // namespace ifx

using namespace ifx;

extern "C" int ifx_t5_attention(const ifx_bf16* q, int32_t ldq, const ifx_bf16* k, int32_t ldk, const ifx_bf16* v, int32_t ldv,
                                ifx_bf16* out, int32_t ldo, const ifx_bf16* rel_bias, const int32_t* seq_lens, int32_t batch,
                                int32_t seq_len_padded, int32_t heads, void* stream) {
  IFX_REQUIRE(q && k && v && out && rel_bias && seq_lens, "ifx_t5_attention: null argument");
  IFX_REQUIRE(batch >= 1 && heads >= 1, "ifx_t5_attention: empty batch / heads");
  IFX_REQUIRE(seq_len_padded >= 32 && seq_len_padded <= 512 && seq_len_padded % 32 == 0,
              "ifx_t5_attention: padded length %d not in [32, 512] step 32 (K and V^T of a head live in LDS)", seq_len_padded);
  IFX_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "ifx_t5_attention: row strides must be multiples of 8");
  t5::AttnArgs a;
  a.q = q;
  a.k = k;
  a.v = v;
  a.out = out;
  a.bias = rel_bias;
  a.seq_lens = seq_lens;
  a.ldq = ldq;
  a.ldk = ldk;
  a.ldv = ldv;
  a.ldo = ldo;
  a.L = seq_len_padded;
  a.heads = heads;
  a.q_tiles = (seq_len_padded + 127) / 128;
  const int L = seq_len_padded;
  const int lds = L * 128 + t5::HD * (L * 2 + 8) + ((2 * L - 1) * 2 + 15) / 16 * 16;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)t5::t5_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(t5::t5_attention_kernel, dim3(batch * heads * a.q_tiles), dim3(256), lds, (hipStream_t)stream, a);
  return check_launch("ifx_t5_attention");
}

extern "C" int ifx_t5_gated_gelu(const ifx_bf16* gate_fc1, ifx_bf16* h, int32_t rows, int32_t ffn, void* stream) {
  IFX_REQUIRE(gate_fc1 && h && rows >= 0 && ffn > 0 && ffn % 8 == 0, "ifx_t5_gated_gelu: bad arguments (ffn %d)", ffn);
  if (rows == 0) return IFX_OK;
  const long long total8 = (long long)rows * (ffn / 8);
  hipLaunchKernelGGL(t5::gated_gelu_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gate_fc1,
                     h, total8, ffn / 8);
  return check_launch("ifx_t5_gated_gelu");
}
