// bwd_pool.hip -- backward of the LAST layer of a set-abstraction stack (1x1 conv -> train-mode BatchNorm -> ReLU -> max over the ns
// neighbours, pointnet_util.py:201-205) WITHOUT its pre-BatchNorm output Y.
//
// The generic pooled backward (bwd_fused3.hip, GMODE 2) rebuilds  dY = gs*G + q*Y + p  per element (G: the pooled gradient at the winner
// rows, zero elsewhere) and therefore reads Y - for SA1's last layer 537 MB of the 1.07 GB the launch moves, and the larger half of its
// staging work.  But Y = A W^T + b is a linear function of the layer's input A = relu(bn(X)) that the kernel stages anyway, so the dense
// part of dY factors through two small matrices:
//
//     dX  = dY W          = A (W^T diag(q) W)  +  (q*b + p) W  +  (gs*G) W          =  A Q + r + Gs W
//     dW  = dY^T A        = Gs^T A  +  diag(q) (W (A^T A) + b (1^T A))  +  p (1^T A)
//
// Q [Ci x Ci] and r [Ci] are computed once per workgroup; per tile (= ONE group of ns = 64 rows) the kernel runs three matrix products on
// the bf16 pipe with the usual three-way split (A Q, Gs W, the Gram matrix A^T A: 2 MFLOP fp32-equivalent, what the generic kernel spends
// on dY W and dY^T A), gathers the <= Co winner rows of A for Gs^T A on the VALU in fp32, and accumulates 1^T A; a small second kernel
// assembles dW in fp64.  Y is neither read here nor - when this route is taken - written by the forward (the pooled forward keeps the
// per-group extremes and the BatchNorm sums in its epilogue).  The gradient that reaches the layer below (dX, with the ReLU +
// BatchNorm-backward sums of that layer reduced on the fly) is the same quantity the generic kernel produces.
//
// Shapes: Co = 128, Ci = 64, ns = 64, X dense (ldx = Ci), M = G * 64: SA1's last layer (1,048,576 rows at B = 32).  Everything else takes
// the generic kernel.
#include "common.h"

#define PA_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

typedef float pa_f32x16 __attribute__((ext_vector_type(16)));
typedef float pa_v4f __attribute__((ext_vector_type(4)));
typedef float pa_v2f __attribute__((ext_vector_type(2)));
typedef __bf16 pa_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pa_bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t pa_v2u __attribute__((ext_vector_type(2)));
typedef uint32_t pa_v4u __attribute__((ext_vector_type(4)));

struct PoolAlgArgs {
    const float *dout; int lddo;          // pooled gradient [G, Co]
    const float *ywin;                    // winners' pre-BatchNorm values [G, Co]
    const int32_t *arg;                   // winners' rows within their group [G, Co]
    const float *coef;                    // [5][Co]: scale, shift, gs, q, p of THIS layer's BatchNorm backward
    const float *x; int ldx;              // pre-BatchNorm output of the layer below [M, Ci]
    const float *in_scale, *in_shift;     // its BatchNorm affine: A = relu(in_scale * x + in_shift)
    const float *w; int ldw;              // [Co, Ci]
    const float *bias;                    // [Co]
    float *dx; int lddx;                  // [M, Ci]
    const float *pstat;                   // [4][Ci]: scale, shift, mean, invstd of the layer below
    double *partials;                     // [P2C_STAT_SLOTS][2][Ci]
    float *wgacc;                         // [gridDim.x][Co*Ci + Ci*Ci + Ci]: Gs^T A | A^T A | 1^T A of every workgroup (plain stores)
    const float *qr;                      // [Ci*Ci + Ci]: Q = W^T diag(q) W and r = (q*b + p) W (pool_alg_prep_kernel)
    int G;
};

__device__ __forceinline__ uint32_t pa_pk(float a, float b)
{
    const pa_v2f v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pa_bf16x2));
}
__device__ __forceinline__ float pa_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float pa_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }
struct PaPair { uint32_t ph, pm, pl; float h0, h1, m0, m1, l0, l1; };
__device__ __forceinline__ PaPair pa_split2(float x0, float x1)            // x = h + m + l, h and m exactly representable in bf16
{
    PaPair r;
    r.ph = pa_pk(x0, x1);
    r.h0 = pa_lo(r.ph); r.h1 = pa_hi(r.ph);
    const float r0 = x0 - r.h0, r1 = x1 - r.h1;
    r.pm = pa_pk(r0, r1);
    r.m0 = pa_lo(r.pm); r.m1 = pa_hi(r.pm);
    r.l0 = r0 - r.m0; r.l1 = r1 - r.m1;
    r.pl = pa_pk(r.l0, r.l1);
    return r;
}

#ifdef PA_TRACE        // tools/bench_pool_alg.py --trace: shader-clock time of every phase, summed over the tiles of workgroup 0 / wave 0
__device__ unsigned long long pa_trace_buf[12];
extern "C" int p2c_pool_alg_trace_read(void *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(pa_trace_buf), sizeof(pa_trace_buf)) == hipSuccess ? 0 : 1; }
#define PA_T(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); tr[i] += n_ - tl; tl = n_; } while (0)
#else
#define PA_T(i) do { } while (0)
#endif

#define PA_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C_, 0, 0, 0)
// six of the nine piece products, smallest first (piece index 0 = hi, 1 = mid, 2 = lo)
#define PA_SIX(A_, B_, C_)                                                                                         \
    do {                                                                                                           \
        PA_MFMA(A_[1], B_[1], C_); PA_MFMA(A_[0], B_[2], C_); PA_MFMA(A_[2], B_[0], C_);                           \
        PA_MFMA(A_[0], B_[1], C_); PA_MFMA(A_[1], B_[0], C_); PA_MFMA(A_[0], B_[0], C_);                           \
    } while (0)

template <int Co, int Ci>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) pool_alg_bwd_kernel(PoolAlgArgs a)
{
    constexpr int BM = 64;                                                 // one tile = one group of 64 neighbours
    constexpr int LDA = 2 * Ci + 16, PLA = BM * LDA;                       // A row-major pieces  [3][BM][LDA]   (k = ci contiguous)
    constexpr int LDT = 2 * BM + 16, PLT = Ci * LDT;                       // A transposed pieces [3][Ci][LDT]   (k = rows contiguous)
    constexpr int LDG = 2 * Co + 16, PLG = BM * LDG;                       // Gs row-major pieces [3][BM][LDG]   (k = co contiguous)
    constexpr int LDF = Ci + 8;                                            // fp32 rows: raw x, and A (4 rows = 288 dwords = 32 mod 64: the epilogue's two half-waves hit disjoint banks)
    constexpr int NQG = Co / 16, NQA = Ci / 16, NQR = BM / 16;
    constexpr int ACCN = Co * Ci + Ci * Ci + Ci;
    static_assert(Ci == 64 && Co == 128, "instantiated for SA1's last layer");
    extern __shared__ __attribute__((aligned(16))) unsigned char pa_smem[];
    unsigned char *AR = pa_smem;
    unsigned char *AT = AR + 3 * PLA;
    unsigned char *GR = AT + 3 * PLT;
    float *XF = reinterpret_cast<float *>(GR + 3 * PLG);                   // raw x   [BM][LDF]
    float *AF = XF + BM * LDF;                                             // A fp32  [BM][LDF]
    float *red = AF + BM * LDF;                                            // [Ci] column sums at the end

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef PA_TRACE
    const unsigned long long t_start = __builtin_readcyclecounter();
#endif
    const int l31 = lane & 31, lh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;                               // 2 x 2 wave grid over the 64 x 64 outputs (dX and Gram alike)
    const int nk = (a.G - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto tile_of = [&](int k) { return (int)blockIdx.x + k * (int)gridDim.x; };
    // staging unit of the thread: 4 rows x 4 channels; 8 consecutive lanes along a row (whole 128-byte lines per quarter wave)
    const int c4 = (tid & 7) + 8 * (tid >> 7), rg = (tid >> 3) & 15;
    // winners: thread pair (2c, 2c+1) handles output channel c; each of the two takes 32 input channels of the gathered row
    const int gc = tid >> 1, ghalf = tid & 1;

    // ---------------- prefetch of the first tile
    pa_v4f rx[4];
    float rdo, ryw;
    int rar;
    auto gload = [&](int t) {
        const size_t m0 = (size_t)t * BM;
#pragma unroll
        for (int j = 0; j < 4; ++j) rx[j] = *reinterpret_cast<const pa_v4f *>(a.x + (m0 + 4 * rg + j) * a.ldx + 4 * c4);
        rdo = a.dout[(size_t)t * a.lddo + gc];
        ryw = a.ywin[(size_t)t * Co + gc];
        rar = a.arg[(size_t)t * Co + gc];
    };
    if (nk > 0) gload(tile_of(0));

    // this wave's slabs as register-resident bf16 fragments: lane (i, h), k-step s holds B[16 s + 8 h + e][32 wc + i]
    pa_bf16x8 wb[3][NQG], qb[3][NQA];
    {
        const float *wp = a.w + (size_t)(8 * lh) * a.ldw + wc * 32 + l31;
#pragma unroll
        for (int s = 0; s < NQG; ++s) {
            pa_v4u ph, pm, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const PaPair sp = pa_split2(wp[(size_t)(16 * s + 2 * e) * a.ldw], wp[(size_t)(16 * s + 2 * e + 1) * a.ldw]);
                ph[e] = sp.ph; pm[e] = sp.pm; pl[e] = sp.pl;
            }
            wb[0][s] = __builtin_bit_cast(pa_bf16x8, ph); wb[1][s] = __builtin_bit_cast(pa_bf16x8, pm); wb[2][s] = __builtin_bit_cast(pa_bf16x8, pl);
        }
        const float *qp = a.qr + (8 * lh) * Ci + wc * 32 + l31;        // Q and r: computed once by pool_alg_prep_kernel (in-kernel, a chain of
        // 128 dependent L2 round trips per workgroup, it took 46 k cycles = 21 us of a 380 us launch)
#pragma unroll
        for (int s = 0; s < NQA; ++s) {
            pa_v4u ph, pm, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const PaPair sp = pa_split2(qp[(16 * s + 2 * e) * Ci], qp[(16 * s + 2 * e + 1) * Ci]);
                ph[e] = sp.ph; pm[e] = sp.pm; pl[e] = sp.pl;
            }
            qb[0][s] = __builtin_bit_cast(pa_bf16x8, ph); qb[1][s] = __builtin_bit_cast(pa_bf16x8, pm); qb[2][s] = __builtin_bit_cast(pa_bf16x8, pl);
        }
    }
    const int xcol = wc * 32 + l31;
    const float rconst = a.qr[Ci * Ci + xcol];
    for (int i = tid; i < 3 * PLG / 16; i += 256) reinterpret_cast<pa_v4u *>(GR)[i] = pa_v4u{0u, 0u, 0u, 0u};      // the Gs image starts empty
    // per-thread constants
    const pa_v4f isc = *reinterpret_cast<const pa_v4f *>(a.in_scale + 4 * c4), ish = *reinterpret_cast<const pa_v4f *>(a.in_shift + 4 * c4);
    const float c_sc = a.coef[gc], c_sh = a.coef[Co + gc], c_gs = a.coef[2 * Co + gc];
    const float psc = a.pstat[xcol], psh = a.pstat[Ci + xcol], pmu = a.pstat[2 * Ci + xcol], pis = a.pstat[3 * Ci + xcol];
    const float npm = -pmu * pis;
    __syncthreads();

    // ---------------- accumulators that live for the whole kernel
    pa_f32x16 accG0, accG1;                                                // Gram block (32 wr.., 32 wc..)
#pragma unroll
    for (int r = 0; r < 16; ++r) accG0[r] = accG1[r] = 0.f;
    float ga[32];                                                          // (Gs^T A)[gc][32 ghalf ..]
#pragma unroll
    for (int e = 0; e < 32; ++e) ga[e] = 0.f;
    pa_v4f csum = {0.f, 0.f, 0.f, 0.f};                                    // 1^T A of the unit's 4 channels
    double s1 = 0.0, s2 = 0.0;
    int prev_row = 0;                                                      // where this thread's Gs entry of the previous tile sits

    const int a_row = (32 * wr + l31) * LDA + 16 * lh;                     // A Q  : A operand, row of the tile, k = ci
    const int g_row = (32 * wr + l31) * LDG + 16 * lh;                     // Gs W : A operand, row of the tile, k = co
    const int ta_row = (32 * wr + l31) * LDT + 16 * lh;                    // Gram : A operand = transposed row ci (output row block wr)
    const int tb_row = (32 * wc + l31) * LDT + 16 * lh;                    //        B operand = transposed row ci' (output column block wc)

#ifdef PA_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_readcyclecounter();
    const unsigned long long t_loop0 = tl, rt_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int k = 0; k < nk; ++k) {
        const int t = tile_of(k);
        const size_t m0 = (size_t)t * BM;
        PA_T(7);
        // ================= stage: A = relu(bn(x)) in both piece layouts, raw x and A in fp32; this tile's winners into the Gs image =================
        {
            pa_v4f av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const pa_v4f v = rx[j];
                pa_v4f o;
                o.x = fmaxf(isc.x * v.x + ish.x, 0.f); o.y = fmaxf(isc.y * v.y + ish.y, 0.f);
                o.z = fmaxf(isc.z * v.z + ish.z, 0.f); o.w = fmaxf(isc.w * v.w + ish.w, 0.f);
                av[j] = o;
                csum += o;
                *reinterpret_cast<pa_v4f *>(XF + (4 * rg + j) * LDF + 4 * c4) = v;
                *reinterpret_cast<pa_v4f *>(AF + (4 * rg + j) * LDF + 4 * c4) = o;
            }
            float H[4][4], Mi[4][4], L[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                                  // pairs along the channels: the packed dwords ARE the row-major pieces
                const PaPair p0 = pa_split2(av[j].x, av[j].y), p1 = pa_split2(av[j].z, av[j].w);
                H[j][0] = p0.h0; H[j][1] = p0.h1; H[j][2] = p1.h0; H[j][3] = p1.h1;
                Mi[j][0] = p0.m0; Mi[j][1] = p0.m1; Mi[j][2] = p1.m0; Mi[j][3] = p1.m1;
                L[j][0] = p0.l0; L[j][1] = p0.l1; L[j][2] = p1.l0; L[j][3] = p1.l1;
                unsigned char *d = AR + (4 * rg + j) * LDA + 8 * c4;
                *reinterpret_cast<pa_v2u *>(d) = pa_v2u{p0.ph, p1.ph};
                *reinterpret_cast<pa_v2u *>(d + PLA) = pa_v2u{p0.pm, p1.pm};
                *reinterpret_cast<pa_v2u *>(d + 2 * PLA) = pa_v2u{p0.pl, p1.pl};
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {                                  // pairs along the rows: 4 rows of one channel = 8 bytes per piece
                unsigned char *d = AT + (4 * c4 + e) * LDT + 8 * rg;
                *reinterpret_cast<pa_v2u *>(d) = pa_v2u{pa_pk(H[0][e], H[1][e]), pa_pk(H[2][e], H[3][e])};
                *reinterpret_cast<pa_v2u *>(d + PLT) = pa_v2u{pa_pk(Mi[0][e], Mi[1][e]), pa_pk(Mi[2][e], Mi[3][e])};
                *reinterpret_cast<pa_v2u *>(d + 2 * PLT) = pa_v2u{pa_pk(L[0][e], L[1][e]), pa_pk(L[2][e], L[3][e])};
            }
        }
        // the winner of channel gc: gs * (pooled gradient where the pooled value passed its ReLU - the forward's own two roundings)
        const float cg = c_gs * ((c_sc * ryw + c_sh > 0.f) ? rdo : 0.f);
        const int grow = rar;
        if (ghalf == 0) {
            const PaPair sp = pa_split2(cg, 0.f);
            unsigned char *o = GR + prev_row * LDG + 2 * gc, *n = GR + grow * LDG + 2 * gc;
            *reinterpret_cast<uint16_t *>(o) = 0; *reinterpret_cast<uint16_t *>(o + PLG) = 0; *reinterpret_cast<uint16_t *>(o + 2 * PLG) = 0;
            *reinterpret_cast<uint16_t *>(n) = (uint16_t)(sp.ph & 0xFFFFu);
            *reinterpret_cast<uint16_t *>(n + PLG) = (uint16_t)(sp.pm & 0xFFFFu);
            *reinterpret_cast<uint16_t *>(n + 2 * PLG) = (uint16_t)(sp.pl & 0xFFFFu);
            prev_row = grow;
        }
        PA_T(0);
        PA_LDS_BARRIER();
        PA_T(1);
        // ================= the next tile's rows: in flight under the matrix products =================
        if (k + 1 < nk) gload(tile_of(k + 1));
        __builtin_amdgcn_sched_barrier(0);
        // ================= dX block = Gs W + A Q  (two accumulators: consecutive MFMAs never share one) =================
        pa_f32x16 accX0, accX1;
#pragma unroll
        for (int r = 0; r < 16; ++r) accX0[r] = accX1[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NQG; s += 2) {
            pa_bf16x8 f0[3], f1[3], b0[3], b1[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                f0[p] = *reinterpret_cast<const pa_bf16x8 *>(GR + g_row + p * PLG + 32 * s);
                f1[p] = *reinterpret_cast<const pa_bf16x8 *>(GR + g_row + p * PLG + 32 * (s + 1));
                b0[p] = wb[p][s]; b1[p] = wb[p][s + 1];
            }
            PA_MFMA(f0[1], b0[1], accX0); PA_MFMA(f1[1], b1[1], accX1);
            PA_MFMA(f0[0], b0[2], accX0); PA_MFMA(f1[0], b1[2], accX1);
            PA_MFMA(f0[2], b0[0], accX0); PA_MFMA(f1[2], b1[0], accX1);
            PA_MFMA(f0[0], b0[1], accX0); PA_MFMA(f1[0], b1[1], accX1);
            PA_MFMA(f0[1], b0[0], accX0); PA_MFMA(f1[1], b1[0], accX1);
            PA_MFMA(f0[0], b0[0], accX0); PA_MFMA(f1[0], b1[0], accX1);
        }
#pragma unroll
        for (int s = 0; s < NQA; s += 2) {
            pa_bf16x8 f0[3], f1[3], b0[3], b1[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                f0[p] = *reinterpret_cast<const pa_bf16x8 *>(AR + a_row + p * PLA + 32 * s);
                f1[p] = *reinterpret_cast<const pa_bf16x8 *>(AR + a_row + p * PLA + 32 * (s + 1));
                b0[p] = qb[p][s]; b1[p] = qb[p][s + 1];
            }
            PA_MFMA(f0[1], b0[1], accX0); PA_MFMA(f1[1], b1[1], accX1);
            PA_MFMA(f0[0], b0[2], accX0); PA_MFMA(f1[0], b1[2], accX1);
            PA_MFMA(f0[2], b0[0], accX0); PA_MFMA(f1[2], b1[0], accX1);
            PA_MFMA(f0[0], b0[1], accX0); PA_MFMA(f1[0], b1[1], accX1);
            PA_MFMA(f0[1], b0[0], accX0); PA_MFMA(f1[1], b1[0], accX1);
            PA_MFMA(f0[0], b0[0], accX0); PA_MFMA(f1[0], b1[0], accX1);
        }
        PA_T(2);
        // ================= Gram block += A^T A (k = the 64 rows) =================
#pragma unroll
        for (int s = 0; s < NQR; s += 2) {
            pa_bf16x8 f0[3], f1[3], b0[3], b1[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                f0[p] = *reinterpret_cast<const pa_bf16x8 *>(AT + ta_row + p * PLT + 32 * s);
                f1[p] = *reinterpret_cast<const pa_bf16x8 *>(AT + ta_row + p * PLT + 32 * (s + 1));
                b0[p] = *reinterpret_cast<const pa_bf16x8 *>(AT + tb_row + p * PLT + 32 * s);
                b1[p] = *reinterpret_cast<const pa_bf16x8 *>(AT + tb_row + p * PLT + 32 * (s + 1));
            }
            PA_MFMA(f0[1], b0[1], accG0); PA_MFMA(f1[1], b1[1], accG1);
            PA_MFMA(f0[0], b0[2], accG0); PA_MFMA(f1[0], b1[2], accG1);
            PA_MFMA(f0[2], b0[0], accG0); PA_MFMA(f1[2], b1[0], accG1);
            PA_MFMA(f0[0], b0[1], accG0); PA_MFMA(f1[0], b1[1], accG1);
            PA_MFMA(f0[1], b0[0], accG0); PA_MFMA(f1[1], b1[0], accG1);
            PA_MFMA(f0[0], b0[0], accG0); PA_MFMA(f1[0], b1[0], accG1);
        }
        PA_T(3);
        // ================= Gs^T A: the winner's row of A, in fp32 =================
        {
            const float *ap = AF + grow * LDF + 32 * ghalf;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const pa_v4f v = *reinterpret_cast<const pa_v4f *>(ap + 4 * q);
                ga[4 * q + 0] = __builtin_fmaf(cg, v.x, ga[4 * q + 0]); ga[4 * q + 1] = __builtin_fmaf(cg, v.y, ga[4 * q + 1]);
                ga[4 * q + 2] = __builtin_fmaf(cg, v.z, ga[4 * q + 2]); ga[4 * q + 3] = __builtin_fmaf(cg, v.w, ga[4 * q + 3]);
            }
        }
        PA_T(4);
        // ================= epilogue: + r, the ReLU + BatchNorm-backward sums of the layer below, dX stores =================
        {
            float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
            float vx[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float xr = XF[row * LDF + xcol];
                const float v = (accX0[r] + accX1[r]) + rconst;
                vx[r] = v;
                const float g = (psc * xr + psh > 0.f) ? v : 0.f;          // the forward's two roundings
                t1[r & 3] += g;
                t2[r & 3] = __builtin_fmaf(g, __builtin_fmaf(xr, pis, npm), t2[r & 3]);
            }
            s1 += (double)((t1[0] + t1[1]) + (t1[2] + t1[3]));
            s2 += (double)((t2[0] + t2[1]) + (t2[2] + t2[3]));
            asm volatile("" ::"v"(rx[3]));                                 // settle the prefetch while only loads are outstanding (vmcnt is in-order)
            const uint32_t dxo = (uint32_t)((32 * wr + 4 * lh) * a.lddx + xcol) * 4u;      // per-lane byte offset inside the tile; the rows are scalar
#pragma unroll
            for (int r = 0; r < 16; ++r)
                *reinterpret_cast<float *>(reinterpret_cast<char *>(a.dx + (m0 + (r & 3) + 8 * (r >> 2)) * a.lddx) + dxo) = vx[r];
        }
        PA_T(5);
        PA_LDS_BARRIER();                                                  // every read of this tile's LDS image is done
        PA_T(6);
    }
#ifdef PA_TRACE
    const unsigned long long t_loop1 = __builtin_readcyclecounter(), rt_loop1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 8; ++i) pa_trace_buf[i] = tr[i];
        pa_trace_buf[8] = t_loop0 - t_start; pa_trace_buf[9] = t_loop1 - t_loop0; pa_trace_buf[10] = rt_loop1 - rt_loop0;
    }
#endif

    // ---------------- flush: this workgroup's partial sums as plain coalesced stores into its own row of the workspace (summed in fp64 by
    // pool_alg_reduce_kernel).  fp64 atomics into 8 shared copies - 12,352 per workgroup - took 224 k cycles = 100 us of a 380 us launch.
    float *wg = a.wgacc + (size_t)blockIdx.x * ACCN;
#pragma unroll
    for (int q = 0; q < 8; ++q)
        *reinterpret_cast<pa_v4f *>(wg + gc * Ci + 32 * ghalf + 4 * q) = pa_v4f{ga[4 * q], ga[4 * q + 1], ga[4 * q + 2], ga[4 * q + 3]};
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * lh;
        wg[Co * Ci + row * Ci + xcol] = accG0[r] + accG1[r];
    }
    for (int i = tid; i < Ci; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(&red[4 * c4 + e], csum[e]);
    __syncthreads();
    if (tid < Ci) wg[Co * Ci + Ci * Ci + tid] = red[tid];
    {
        const double u1 = s1 + __shfl_xor(s1, 32), u2 = s2 + __shfl_xor(s2, 32);
        if (lh == 0) {
            double *o = a.partials + (size_t)(blockIdx.x % P2C_STAT_SLOTS) * 2 * Ci;
            atomicAdd(&o[xcol], u1);
            atomicAdd(&o[Ci + xcol], u2);
        }
    }
#ifdef PA_TRACE
    if (blockIdx.x == 0 && tid == 0) pa_trace_buf[11] = __builtin_readcyclecounter() - t_loop1;
#endif
}

// Q [Ci, Ci] = W^T diag(q) W and r [Ci] = (q*b + p) W.  W goes through LDS first (one pipelined burst of coalesced loads: with one
// thread per output reading W from global memory the kernel was a chain of 128 L2 round trips - 30 us)
__global__ void __launch_bounds__(256) pool_alg_prep_kernel(const float *__restrict__ coef, const float *__restrict__ w, int ldw,
                                                            const float *__restrict__ bias, float *__restrict__ qr, int Co, int Ci)
{
    extern __shared__ __attribute__((aligned(16))) float pw[];             // [Co][Ci] W, then [Co] q, [Co] q*b + p
    float *qs = pw + Co * Ci, *ts = qs + Co;
    for (int i = threadIdx.x; i < Co * Ci / 4; i += 256) {
        const int c = i / (Ci / 4), k4 = i - c * (Ci / 4);
        *reinterpret_cast<pa_v4f *>(pw + c * Ci + 4 * k4) = *reinterpret_cast<const pa_v4f *>(w + (size_t)c * ldw + 4 * k4);
    }
    for (int c = threadIdx.x; c < Co; c += 256) {
        qs[c] = coef[3 * Co + c];
        ts[c] = __builtin_fmaf(coef[3 * Co + c], bias[c], coef[4 * Co + c]);
    }
    __syncthreads();
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o < Ci * Ci) {
        const int k = o / Ci, n = o - k * Ci;
        float s = 0.f;
        for (int c = 0; c < Co; ++c) s = __builtin_fmaf(qs[c] * pw[c * Ci + k], pw[c * Ci + n], s);
        qr[o] = s;
    } else if (o < Ci * Ci + Ci) {
        const int n = o - Ci * Ci;
        float s = 0.f;
        for (int c = 0; c < Co; ++c) s = __builtin_fmaf(ts[c], pw[c * Ci + n], s);
        qr[o] = s;
    }
}

// the workgroups' partial sums -> fp64 totals.  Block = 32 elements x 8 row slices (a row's 32 elements are one 128-byte line), 386 blocks
__global__ void __launch_bounds__(256) pool_alg_reduce_kernel(const float *__restrict__ wgacc, int nwg, int accn, double *__restrict__ tot)
{
    __shared__ double part[8][32];
    const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    double s0 = 0.0, s1 = 0.0;
    if (i < accn) {
        int g = sl;
#pragma unroll 4
        for (; g + 8 < nwg; g += 16) {
            s0 += (double)wgacc[(size_t)g * accn + i];
            s1 += (double)wgacc[(size_t)(g + 8) * accn + i];
        }
        if (g < nwg) s0 += (double)wgacc[(size_t)g * accn + i];
    }
    part[sl][e] = s0 + s1;
    __syncthreads();
    if (sl == 0 && i < accn) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][e];
        tot[i] = t;
    }
}

// dW [Co, Ci] = Gs^T A + diag(q) (W (A^T A) + b (1^T A)) + p (1^T A), in fp64; the Gram totals through LDS (a block = 4 output channels)
__global__ void __launch_bounds__(256) pool_alg_finalize_kernel(const double *__restrict__ tot, const float *__restrict__ coef,
                                                               const float *__restrict__ w, int ldw, const float *__restrict__ bias,
                                                               float *__restrict__ dw, int lddw, int Co, int Ci)
{
    extern __shared__ __attribute__((aligned(16))) double pg[];            // [Ci][Ci] Gram, [Ci] column sums
    for (int i = threadIdx.x; i < Ci * Ci + Ci; i += 256) pg[i] = tot[Co * Ci + i];
    __syncthreads();
    const int c = blockIdx.x * (256 / Ci) + threadIdx.x / Ci, n = threadIdx.x % Ci;
    if (c >= Co) return;
    const double gsa = tot[c * Ci + n], cs = pg[Ci * Ci + n];
    double dot = 0.0;
    for (int k = 0; k < Ci; ++k) dot += (double)w[(size_t)c * ldw + k] * pg[k * Ci + n];
    const double q = coef[3 * Co + c], p = coef[4 * Co + c];
    dw[(size_t)c * lddw + n] = (float)(gsa + q * (dot + (double)bias[c] * cs) + p * cs);
}

extern "C" int p2c_linear_bwd_pool_alg_supported(int M, int Co, int Ci, int ns)
{
    return (Co == 128 && Ci == 64 && ns == 64 && M >= 64 * 256 && M % 64 == 0) ? 1 : 0;
}

// workspace (no initialisation needed): 256 rows of per-workgroup partial sums (fp32) | their fp64 totals | Q and r
extern "C" size_t p2c_linear_bwd_pool_alg_ws_bytes(int Co, int Ci)
{
    const size_t accn = (size_t)Co * Ci + (size_t)Ci * Ci + Ci;
    return 256 * accn * sizeof(float) + accn * sizeof(double) + ((size_t)Ci * Ci + Ci) * sizeof(float) + 64;
}

extern "C" int p2c_linear_bwd_pool_alg_f32(const float *dout, int lddo, const float *ywin, const int32_t *arg, const float *coef, const float *X,
                                           int ldx, const float *in_scale, const float *in_shift, const float *W, int ldw, const float *bias,
                                           float *dX, int lddx, const float *prev_stat, double *bwd_partials, void *ws, float *dW, int lddw,
                                           int M, int Co, int Ci, int ns, void *stream)
{
    if (!dout || !ywin || !arg || !coef || !X || !in_scale || !in_shift || !W || !bias || !dX || !prev_stat || !bwd_partials || !ws || !dW)
        return P2C_EINVAL;
    if (!p2c_linear_bwd_pool_alg_supported(M, Co, Ci, ns) || ldx % 4 || lddx < Ci || ldw < Ci || lddo < Co || ((uintptr_t)ws & 15)) return P2C_EINVAL;
    const int accn = Co * Ci + Ci * Ci + Ci;
    const int G = M / ns, grid = G < 256 ? G : 256;
    float *wgacc = (float *)ws;
    double *tot = (double *)(wgacc + (size_t)256 * accn);
    float *qr = (float *)(tot + accn);
    PoolAlgArgs a{dout, lddo, ywin, arg, coef, X, ldx, in_scale, in_shift, W, ldw, bias, dX, lddx, prev_stat, bwd_partials, wgacc, qr, G};
    hipStream_t s = (hipStream_t)stream;
    constexpr int CO = 128, CI = 64, BM = 64;
    constexpr size_t lds = 3 * BM * (2 * CI + 16) + 3 * CI * (2 * BM + 16) + 3 * BM * (2 * CO + 16) + 2 * BM * (CI + 8) * 4 + CI * 4;
    static_assert(lds <= 160 * 1024, "LDS");
    hipLaunchKernelGGL(pool_alg_prep_kernel, dim3(p2c_cdiv(Ci * Ci + Ci, 256)), dim3(256), (size_t)(Co * Ci + 2 * Co) * sizeof(float), s, coef, W, ldw, bias, qr, Co, Ci);
    (void)hipFuncSetAttribute((const void *)pool_alg_bwd_kernel<CO, CI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((pool_alg_bwd_kernel<CO, CI>), dim3(grid), dim3(256), lds, s, a);
    hipLaunchKernelGGL(pool_alg_reduce_kernel, dim3(p2c_cdiv(accn, 32)), dim3(256), 0, s, wgacc, grid, accn, tot);
    hipLaunchKernelGGL(pool_alg_finalize_kernel, dim3(p2c_cdiv(Co, 256 / Ci)), dim3(256), (size_t)(Ci * Ci + Ci) * sizeof(double), s, tot, coef, W, ldw, bias, dW, lddw, Co, Ci);
    P2C_LAUNCH_CHECK();
    return P2C_OK;
}
