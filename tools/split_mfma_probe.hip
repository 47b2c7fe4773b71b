// tools/split_mfma_probe.hip -- VERDICT r4 item 6: the regressor's 128 -> 128 ReLU layer (csrc/mlp.hip, regress_tail_kernel /
// linear_wide_kernel run at 0.77 - 0.81 of the fp32 MFMA peak, i.e. the PEAK is their ceiling) on the bf16 matrix pipe
// with every fp32 operand split EXACTLY into three bf16 terms (x = x1 + x2 + x3, 8 + 8 + 8 mantissa bits) and the six
// significant cross products (x1 w1, x1 w2, x2 w1, x2 w2, x1 w3, x3 w1; dropped: 2^-24 relative and below) accumulated
// in fp32 by v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate, and not on the VALU datapath).
// Prints, for M rows: time per launch / per 16 rows of (a) the fp32 16x16x4 form and (b) the split form INCLUDING the
// split's VALU work, and the max / rms error of both against an fp64 evaluation of the same fp32 inputs.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/split_mfma_probe.hip -o tools/split_mfma_probe && ./tools/split_mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
constexpr int C = 128;
constexpr int SF = 132;          // fp32 LDS row stride
constexpr int SB = 136;          // bf16 LDS row stride (16-byte aligned, skewed)

__device__ __forceinline__ v4f ld4(const float *p) { return *(const v4f *)p; }

// ---- (a) fp32 operands, v_mfma_f32_16x16x4_f32: the product kernel's layer ------------------------------------------
template <int REPS>
__global__ __launch_bounds__(512) void layer_f32(const float *X, const float *W, const float *B, float *Y, long m)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *w = lds, *b = lds + C * SF;
    for (int i = threadIdx.x; i < C * C; i += blockDim.x)
        w[(i >> 7) * SF + (i & 127)] = W[i];
    for (int i = threadIdx.x; i < C; i += blockDim.x)
        b[i] = B[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int pt = lane & 15, q = lane >> 4;
    const long ntiles = (m + 15) >> 4;
    for (long tile = (long)blockIdx.x * nw + wave; tile < ntiles; tile += (long)gridDim.x * nw) {
        const long row = tile * 16 + pt;
        v4f h0[C / 16], h1[C / 16];
#pragma unroll
        for (int s = 0; s < C / 16; ++s)
            h0[s] = ld4(X + row * C + 16 * s + 4 * q);
        for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll
        for (int t = 0; t < C / 16; ++t)
            h1[t] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < C / 16; ++s) {
            v4f wv[C / 16];
#pragma unroll
            for (int t = 0; t < C / 16; ++t)
                wv[t] = ld4(w + (16 * t + pt) * SF + 16 * s + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int t = 0; t < C / 16; ++t)
                    h1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t][k], h0[s][k], h1[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < C / 16; ++t) {
            v4f v = h1[t] + ld4(b + 16 * t + 4 * q);
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            h0[t] = v;                  // the next layer's input, register to register (as regress_tail_kernel chains its layers)
        }
        }
#pragma unroll
        for (int t = 0; t < C / 16; ++t)
            *(v4f *)(Y + row * C + 16 * t + 4 * q) = h0[t];
    }
}

// ---- (b) three bf16 terms per operand, six products on v_mfma_f32_16x16x32_bf16 -------------------------------------
// x = t1 + t2 + t3 exactly: t1 = bf16(x), t2 = bf16(x - t1), t3 = bf16(x - t1 - t2) (round to nearest each; the
// residuals are exact fp32 subtractions, and the last one fits 8 bits)
__device__ __forceinline__ void split3(const v4f lo, const v4f hi, bf8 &t1, bf8 &t2, bf8 &t3)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? lo[j] : hi[j - 4];
        const __bf16 a = (__bf16)x;
        const float r1 = x - (float)a;
        const __bf16 bb = (__bf16)r1;
        const float r2 = r1 - (float)bb;
        t1[j] = a; t2[j] = bb; t3[j] = (__bf16)r2;
    }
}

// the eight weights of output row `mrow` for slab pair S in the k-slot order of the activations: channels
// 32 S + {4 q + j, 16 + 4 q + j}
__device__ __forceinline__ bf8 w8(const __bf16 *w, int mrow, int S, int q)
{
    const bf4 lo = *(const bf4 *)(w + mrow * SB + 32 * S + 4 * q), hi = *(const bf4 *)(w + mrow * SB + 32 * S + 16 + 4 * q);
    bf8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = lo[j];
        r[4 + j] = hi[j];
    }
    return r;
}

template <int NT, int REPS>      // NT 16-row tiles per wave step (the weight fragments are read once per step)
__global__ __launch_bounds__(512) void layer_split(const float *X, const float *W, const float *B, float *Y, long m)
{
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    __bf16 *w1 = (__bf16 *)ldsb, *w2 = w1 + C * SB, *w3 = w2 + C * SB;
    float *b = (float *)(w3 + C * SB);
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
        const float x = W[i];
        const __bf16 a = (__bf16)x;
        const float r1 = x - (float)a;
        const __bf16 bb = (__bf16)r1;
        const int o = (i >> 7) * SB + (i & 127);
        w1[o] = a; w2[o] = bb; w3[o] = (__bf16)(r1 - (float)bb);
    }
    for (int i = threadIdx.x; i < C; i += blockDim.x)
        b[i] = B[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int pt = lane & 15, q = lane >> 4;
    const long nsteps = (m + 16 * NT - 1) / (16 * NT);
    for (long step = (long)blockIdx.x * nw + wave; step < nsteps; step += (long)gridDim.x * nw) {
        bf8 x1[NT][C / 32], x2[NT][C / 32], x3[NT][C / 32];
        v4f acc[NT][C / 16];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const long row = (step * NT + n) * 16 + pt;
#pragma unroll
            for (int S = 0; S < C / 32; ++S)
                split3(ld4(X + row * C + 32 * S + 4 * q), ld4(X + row * C + 32 * S + 16 + 4 * q), x1[n][S], x2[n][S], x3[n][S]);
        }
        for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int t = 0; t < C / 16; ++t)
                acc[n][t] = ld4(b + 16 * t + 4 * q);
#pragma unroll
        for (int S = 0; S < C / 32; ++S) {
#pragma unroll
            for (int t = 0; t < C / 16; ++t) {
                const bf8 a1 = w8(w1, 16 * t + pt, S, q), a2 = w8(w2, 16 * t + pt, S, q), a3 = w8(w3, 16 * t + pt, S, q);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    v4f c = acc[n][t];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, x1[n][S], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x3[n][S], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x2[n][S], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x1[n][S], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x2[n][S], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x1[n][S], c, 0, 0, 0);
                    acc[n][t] = c;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ReLU, and the next layer's operands: tiles 2 S and 2 S + 1 ARE slab pair S in this k-slot order
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int t = 0; t < C / 16; ++t) {
                v4f v = acc[n][t];
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                acc[n][t] = v;
            }
        if (rep + 1 < REPS) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int S = 0; S < C / 32; ++S)
                    split3(acc[n][2 * S], acc[n][2 * S + 1], x1[n][S], x2[n][S], x3[n][S]);
        }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const long row = (step * NT + n) * 16 + pt;
#pragma unroll
            for (int t = 0; t < C / 16; ++t)
                *(v4f *)(Y + row * C + 16 * t + 4 * q) = acc[n][t];
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main()
{
    const long m = 2396160;          // 3840 patches x 312 points x 2 replicas: one level-4 launch of the bench
    std::vector<float> X((size_t)m * C), W(C * C), B(C);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFFFF) / 16777216.f; };
    for (auto &v : W) v = (rnd() * 2.f - 1.f) * 0.153f;              // xavier_uniform(128, 128)
    for (auto &v : B) v = (rnd() * 2.f - 1.f) * 0.05f;
    for (auto &v : X) { const float u = rnd() * 2.f - 0.6f; v = u > 0.f ? u : 0.f; }      // post-ReLU activations
    float *dX, *dW, *dB, *dY;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dB, B.size() * 4));
    CK(hipMalloc(&dY, X.size() * 4));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    const int check = 2048;
    constexpr int REPS = 4;
    // fp64 reference: the layer once, and chained REPS times (every layer's input = the previous layer's fp64 output)
    std::vector<double> ref1((size_t)check * C), refR((size_t)check * C);
    for (int r = 0; r < check; ++r) {
        double in[C], o[C];
        for (int k = 0; k < C; ++k) in[k] = X[(size_t)r * C + k];
        for (int rep = 0; rep < REPS; ++rep) {
            for (int oc = 0; oc < C; ++oc) {
                double acc = B[oc];
                for (int k = 0; k < C; ++k)
                    acc += (double)W[oc * C + k] * in[k];
                o[oc] = acc > 0 ? acc : 0;
            }
            for (int k = 0; k < C; ++k) in[k] = o[k];
            if (rep == 0)
                for (int k = 0; k < C; ++k) ref1[(size_t)r * C + k] = o[k];
        }
        for (int k = 0; k < C; ++k) refR[(size_t)r * C + k] = o[k];
    }
    const size_t lds_f32 = (size_t)(C * SF + C) * 4, lds_b = (size_t)3 * C * SB * 2 + C * 4;
    CK(hipFuncSetAttribute((const void *)layer_f32<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f32));
    CK(hipFuncSetAttribute((const void *)layer_f32<REPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f32));
    CK(hipFuncSetAttribute((const void *)layer_split<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    CK(hipFuncSetAttribute((const void *)layer_split<1, REPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    CK(hipFuncSetAttribute((const void *)layer_split<2, REPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> out((size_t)check * C);
    const char *name[5] = {"fp32 operands, v_mfma_f32_16x16x4_f32, ONE layer (HBM: 1 KB per row)        ",
                           "3 x bf16 split, 6 products, 16x16x32_bf16, ONE layer                        ",
                           "fp32 operands, v_mfma_f32_16x16x4_f32, 4 layers chained in registers        ",
                           "3 x bf16 split, 6 products, 4 layers chained (split in registers), 16 rows  ",
                           "3 x bf16 split, 6 products, 4 layers chained (split in registers), 32 rows  "};
    for (int variant = 0; variant < 5; ++variant) {
        float best = 1e30f;
        const int reps = variant < 2 ? 1 : REPS;
        for (int it = 0; it < 12; ++it) {
            CK(hipEventRecord(e0, 0));
            if (variant == 0) hipLaunchKernelGGL(layer_f32<1>, dim3(256), dim3(512), lds_f32, 0, dX, dW, dB, dY, m);
            else if (variant == 1) hipLaunchKernelGGL((layer_split<1, 1>), dim3(256), dim3(512), lds_b, 0, dX, dW, dB, dY, m);
            else if (variant == 2) hipLaunchKernelGGL(layer_f32<REPS>, dim3(256), dim3(512), lds_f32, 0, dX, dW, dB, dY, m);
            else if (variant == 3) hipLaunchKernelGGL((layer_split<1, REPS>), dim3(256), dim3(512), lds_b, 0, dX, dW, dB, dY, m);
            else hipLaunchKernelGGL((layer_split<2, REPS>), dim3(256), dim3(512), lds_b, 0, dX, dW, dB, dY, m);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2 && ms < best) best = ms;
        }
        CK(hipMemcpy(out.data(), dY, out.size() * 4, hipMemcpyDeviceToHost));
        const std::vector<double> &ref = reps == 1 ? ref1 : refR;
        double emax = 0, esq = 0, ymax = 0;
        for (size_t i = 0; i < out.size(); ++i) {
            const double e = fabs((double)out[i] - ref[i]);
            emax = e > emax ? e : emax; esq += e * e; ymax = ref[i] > ymax ? ref[i] : ymax;
        }
        printf("%s %7.3f ms per launch = %6.2f ns per 16 rows and layer; %6.1f TFLOP/s of the layers' 2*128*128 FLOP per row; "
               "max |err| %.3e  rms %.3e  (outputs up to %.2f)\n", name[variant], best, best * 1e6 / (m / 16.0) / reps,
               2.0 * C * C * m * reps / (best * 1e-3) / 1e12, emax, sqrt(esq / out.size()), ymax);
    }
    return 0;
}
