// Self-attention in exact fp32 for the checking mode (core/model/precise.py; `evaluate.py --fp32`):
//   out = softmax((q * scale) k^T) v  per (batch, head)       reference dinov2/layers/attention.py:54-71
// on the packed fp32 qkv [B*L, 3*heads*64] of the ViT trunk, head_dim 64.  Rounds 1-3 ran this as a host loop over
// (batch, head) of split-bf16 GEMMs + a softmax pass (~860 launches per click); one launch per block here.
//
// Flash form on the f32-input matrix instruction v_mfma_f32_32x32x2_f32 (exact fp32 multiply-adds, 64 cycles per SIMD):
// a wave owns 32 queries (their scaled rows in registers as B operands: lane = query l % 32, d = 2 s + l / 32), the
// workgroup's waves share 32-key K / V tiles staged in LDS as fp32 (the next tile's loads in flight behind the MFMAs).  S^T = K Q^T gives a lane 16 keys of ITS query's
// column, so the running maximum / sum need one exchange with lane ^ 32, and exp(S^T - m) in place IS the B operand of
// O^T += V^T P^T when the 32 keys are contracted in the order the accumulator holds them (step (j, r): lanes 0-31 supply
// key 8 j + r, lanes 32-63 key 8 j + 4 + r; the A operand reads V's rows in the same order) -- no data movement between
// the two products.
#include "isp_common.h"

namespace {

constexpr int KT = 32;        // keys per tile
constexpr int KPITCH = 65;    // floats per K row in LDS: column reads (one d, 32 keys) hit 32 different banks

// NW waves of 32 queries per workgroup: 4 at large batch, 2 when (batch x heads x query blocks) would not fill the chip
// (the click loop's batch of 2: 108 workgroups of 128 queries for 256 CUs).
template <int NW>
__global__ __launch_bounds__(64 * NW) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L, int heads,
                                                                 float scale) {
    constexpr int NT = 64 * NW;
    constexpr int PER = KT * 64 / 4 / NT;  // float4 pieces of K (and of V) per thread and tile
    __shared__ float ks[KT * KPITCH];
    __shared__ float vs[KT * 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = lane & 31, g = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int D = heads * 64;
    const size_t ld = (size_t)3 * D;
    const float* base = qkv + (size_t)b * L * ld + h * 64;
    const int q = blockIdx.x * (32 * NW) + wid * 32 + n;
    const int qc = q < L ? q : L - 1;
    float qreg[32];
#pragma unroll
    for (int s = 0; s < 32; ++s) qreg[s] = base[(size_t)qc * ld + 2 * s + g] * scale;
    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) o0[i] = 0.f, o1[i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // the tile after the one being multiplied waits in registers: its global loads are issued before the MFMAs of the
    // current tile and written to LDS behind them (piece p of a thread: key (tid + p NT) / 16, 4 floats at d = 4 ((tid + p NT) % 16))
    float4 kreg[PER], vreg[PER];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int idx = tid + p * NT, key = idx >> 4, d0 = (idx & 15) * 4;
            const int kk = k0 + key;
            kreg[p] = vreg[p] = make_float4(0, 0, 0, 0);
            if (kk < L) {
                kreg[p] = *reinterpret_cast<const float4*>(base + (size_t)kk * ld + D + d0);
                vreg[p] = *reinterpret_cast<const float4*>(base + (size_t)kk * ld + 2 * D + d0);
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < L; k0 += KT) {
        __syncthreads();  // the previous tile's reads are done
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int idx = tid + p * NT, key = idx >> 4, d0 = (idx & 15) * 4;
            float* kd = ks + key * KPITCH + d0;
            kd[0] = kreg[p].x, kd[1] = kreg[p].y, kd[2] = kreg[p].z, kd[3] = kreg[p].w;
            *reinterpret_cast<float4*>(vs + key * 64 + d0) = vreg[p];
        }
        __syncthreads();
        if (k0 + KT < L) fetch(k0 + KT);
        // S^T[key][query] = sum_d K[key][d] Q[query][d]: A = K (lane: key n, d = 2 s + g), B = Q
        f32x16 sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[n * KPITCH + 2 * s + g], qreg[s], sacc, 0, 0, 0);
        // a lane holds keys k0 + 8 j + 4 g + r (register 4 j + r) of query n
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = k0 + 8 * (i >> 2) + 4 * g + (i & 3);
            if (key >= L) sacc[i] = -INFINITY;
            mx = fmaxf(mx, sacc[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);  // (finite: every tile holds at least one real key)
        const float corr = expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sacc[i] = expf(sacc[i] - m_new);
            psum += sacc[i];
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * corr + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i) o0[i] *= corr, o1[i] *= corr;
        // O^T[d][query] += sum_key V[key][d] P[key][query], keys in accumulator order
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = 8 * (i >> 2) + 4 * g + (i & 3);
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[key * 64 + n], sacc[i], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[key * 64 + 32 + n], sacc[i], o1, 0, 0, 0);
        }
    }
    if (q < L) {
        const float inv = 1.0f / l_run;
        float* op = out + ((size_t)b * L + q) * D + h * 64 + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<float4*>(op + 8 * j) = make_float4(o0[4 * j] * inv, o0[4 * j + 1] * inv, o0[4 * j + 2] * inv, o0[4 * j + 3] * inv);
            *reinterpret_cast<float4*>(op + 32 + 8 * j) = make_float4(o1[4 * j] * inv, o1[4 * j + 1] * inv, o1[4 * j + 2] * inv, o1[4 * j + 3] * inv);
        }
    }
}

}  // namespace

extern "C" int isp_attention_packed_f32(const float* qkv, float* out, int B, int L, int heads, float scale, void* stream) {
    ISP_CHECK_ARG(qkv && out && B > 0 && B <= 65535 && L > 0 && heads > 0 && heads <= 65535);
    ISP_CHECK_ARG(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0);
    if ((long)((L + 127) / 128) * heads * B >= 512)
        attention_f32_kernel<4><<<dim3((L + 127) / 128, heads, B), 256, 0, (hipStream_t)stream>>>(qkv, out, L, heads, scale);
    else
        attention_f32_kernel<1><<<dim3((L + 31) / 32, heads, B), 64, 0, (hipStream_t)stream>>>(qkv, out, L, heads, scale);
    return isp_launch_status();
}
