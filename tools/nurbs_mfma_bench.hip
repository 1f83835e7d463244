// nurbs_mfma_bench.hip - does the NURBS tensor-product contraction belong on the matrix cores?  (north_star: "MFMA used only if
// the batched basis x control-point product proves a dense enough contraction"; round-3 review item 3.)
//
// Stage 2 of the tensor-product scheme for ONE facet of the metric config: nine products S = T B^T with T [Mu = 50, nv = 10] (the
// stage-1 temps: 3 components x {N-row, D-row}) and B [Mv = 50, nv = 10] the column-basis matrix, which is BANDED - q + 1 = 4
// non-zeros per row (S0 and Sv use the N-row temps with Nv / Dv, Su the D-row temps with Nv).
//   scalar : one lane per point, 4 FMAs per output from LDS (the banded form; what nurbs_fwd_kernel does, minus its epilogue)
//   mfma   : v_mfma_f32_16x16x4_f32 on the dense zero-padded matrices: M = 64 (4 tiles) x N = 64 (4 tiles) x K = 12 (3 steps)
// Both run one wave per facet over 4000 facets (the metric field), outputs reduced to a checksum per facet so that the
// measurement is the contraction, not the 32-byte-per-point store.  Reports time and the two checksums' relative difference.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/nurbs_mfma_bench tools/nurbs_mfma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int MU = 50, MV = 50, NV = 10, Q = 3;

struct Facet { float T[MU][NV][6]; float Bn[MV][4], Bd[MV][4]; int span[MV]; };   // span[j] in [Q, NV - 1]

__global__ __launch_bounds__(64) void scalar_kernel(const Facet* __restrict__ facets, float* __restrict__ out)
{
    __shared__ Facet f;
    const Facet& g = facets[blockIdx.x];
    for (int i = threadIdx.x; i < (int)(sizeof(Facet) / 4); i += 64) reinterpret_cast<float*>(&f)[i] = reinterpret_cast<const float*>(&g)[i];
    __syncthreads();
    float sum = 0.f;
    for (int idx = threadIdx.x; idx < MU * MV; idx += 64) {
        const int i = idx / MV, j = idx - i * MV;
        const int sv = f.span[j];
        float s0[3] = {0, 0, 0}, su[3] = {0, 0, 0}, svv[3] = {0, 0, 0};
#pragma unroll
        for (int s = 0; s <= Q; ++s) {
            const float* t = f.T[i][sv - Q + s];
            const float bn = f.Bn[j][s], bd = f.Bd[j][s];
#pragma unroll
            for (int k = 0; k < 3; ++k) { s0[k] = fmaf(bn, t[k], s0[k]); su[k] = fmaf(bn, t[3 + k], su[k]); svv[k] = fmaf(bd, t[k], svv[k]); }
        }
        sum += (s0[0] + s0[1] + s0[2]) + 0.5f * (su[0] + su[1] + su[2]) + 0.25f * (svv[0] + svv[1] + svv[2]);
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if (threadIdx.x == 0) out[blockIdx.x] = sum;
}

typedef float float4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void mfma_kernel(const Facet* __restrict__ facets, float* __restrict__ out)
{
    // dense operands in LDS: A_w [64][12] for w = 0..5 (temps, rows >= MU and k >= NV zero), Bn / Bd dense [12][64]
    __shared__ float A[6][64][12];
    __shared__ float Bn[12][64], Bd[12][64];
    const Facet& g = facets[blockIdx.x];
    for (int i = threadIdx.x; i < 6 * 64 * 12; i += 64) reinterpret_cast<float*>(A)[i] = 0.f;
    for (int i = threadIdx.x; i < 12 * 64; i += 64) { reinterpret_cast<float*>(Bn)[i] = 0.f; reinterpret_cast<float*>(Bd)[i] = 0.f; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < MU * NV * 6; idx += 64) {
        const int i = idx / (NV * 6), r = idx - i * NV * 6, c = r / 6, w = r - c * 6;
        A[w][i][c] = g.T[i][c][w];
    }
    for (int idx = threadIdx.x; idx < MV * 4; idx += 64) {
        const int j = idx >> 2, s = idx & 3;
        Bn[g.span[j] - Q + s][j] = g.Bn[j][s];
        Bd[g.span[j] - Q + s][j] = g.Bd[j][s];
    }
    __syncthreads();
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    float sum = 0.f;
    // nine products: (w, B, weight): S0 = T[0..2] Bn, Su = T[3..5] Bn, Sv = T[0..2] Bd
    for (int prod = 0; prod < 9; ++prod) {
        const int w = prod < 3 ? prod : (prod < 6 ? prod : prod - 6);
        const float (*B)[64] = prod < 6 ? Bn : Bd;
        const float weight = prod < 3 ? 1.0f : (prod < 6 ? 0.5f : 0.25f);
        for (int mt = 0; mt < 4; ++mt)
            for (int nt = 0; nt < 4; ++nt) {
                float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    const float a = A[w][mt * 16 + li][ks * 4 + lk];
                    const float b = B[ks * 4 + lk][nt * 16 + li];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
                }
                sum += weight * ((acc[0] + acc[1]) + (acc[2] + acc[3]));        // (padding rows / columns are zero)
            }
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if (threadIdx.x == 0) out[blockIdx.x] = sum;
}

int main()
{
    const int n = 4000;
    std::vector<Facet> host(n);
    srand(7);
    auto rnd = []() { return (float)rand() / RAND_MAX - 0.5f; };
    for (auto& f : host) {
        for (int i = 0; i < MU; ++i) for (int c = 0; c < NV; ++c) for (int w = 0; w < 6; ++w) f.T[i][c][w] = rnd();
        for (int j = 0; j < MV; ++j) { f.span[j] = Q + (j * (NV - Q)) / MV; for (int s = 0; s < 4; ++s) { f.Bn[j][s] = rnd(); f.Bd[j][s] = rnd(); } }
    }
    Facet* dev; float *o1, *o2;
    hipMalloc(&dev, sizeof(Facet) * n); hipMalloc(&o1, 4 * n); hipMalloc(&o2, 4 * n);
    hipMemcpy(dev, host.data(), sizeof(Facet) * n, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms[2] = {0, 0};
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 3; ++rep) { if (which) mfma_kernel<<<n, 64>>>(dev, o2); else scalar_kernel<<<n, 64>>>(dev, o1); }
        hipEventRecord(a);
        for (int rep = 0; rep < 20; ++rep) { if (which) mfma_kernel<<<n, 64>>>(dev, o2); else scalar_kernel<<<n, 64>>>(dev, o1); }
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms[which], a, b); ms[which] /= 20;
    }
    std::vector<float> h1(n), h2(n);
    hipMemcpy(h1.data(), o1, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), o2, 4 * n, hipMemcpyDeviceToHost);
    double num = 0, den = 0;
    for (int i = 0; i < n; ++i) { num += (double)(h1[i] - h2[i]) * (h1[i] - h2[i]); den += (double)h1[i] * h1[i]; }
    const double macs_banded = (double)n * MU * MV * 4 * 9, macs_dense = (double)n * 64 * 64 * 12 * 9;
    printf("{\"facets\": %d, \"scalar_banded_us\": %.1f, \"mfma_dense_us\": %.1f, \"checksum_rel_diff\": %.2e, "
           "\"macs_banded\": %.3g, \"macs_dense_padded\": %.3g, \"scalar_TFLOPs\": %.2f, \"mfma_TFLOPs_dense\": %.2f, "
           "\"what\": \"stage 2 of the tensor-product NURBS scheme for 4000 facets (50 x 50 points, 10 x 10 nets, degree 3): nine products per facet, "
           "one wave per facet, outputs reduced to a checksum; v_mfma_f32_16x16x4_f32 on zero-padded dense operands vs 4 FMAs per output from LDS\"}\n",
           n, ms[0] * 1e3, ms[1] * 1e3, den > 0 ? sqrt(num / den) : 0.0, macs_banded, macs_dense,
           2 * macs_banded / (ms[0] * 1e-3) / 1e12, 2 * macs_dense / (ms[1] * 1e-3) / 1e12);
    return 0;
}
