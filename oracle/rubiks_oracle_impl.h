/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * Type-generic body of the CPU oracle; included once per scalar type by
 * rubiks_oracle.c with RK_T (scalar type) and RK_FN(name) (symbol suffixing)
 * defined.  See rubiks_oracle.c for the header comment and the parity status.
 *
 * Every function restates the arithmetic of one reference kernel, element for
 * element (same operand order, same int/float conversions), and cites the
 * reference file:line it follows.  "ref3d" = cuda_src/rubiks3d_kernels.cu,
 * "ref2d" = cuda_src/rubiks2d_kernels.cu, "refcpp" = cuda_src/rubiks.cpp.
 */

/* ------------------------------------------------------------------ */
/* shared helpers                                                      */
/* ------------------------------------------------------------------ */

/* floor of the shift the way ref3d:65-70 does it: floorf() -- i.e. through
 * fp32 even when T is double -- then truncated into an int. */
static inline int RK_FN(floor3d)(RK_T s) { return (int)floorf((float)s); }

/* zero-filled tap of a [N,T,C,H,W] tensor (ref3d:102-106 and the 7 siblings) */
static inline RK_T RK_FN(tap5)(const RK_T* p, int n, int t, int c, int h, int w,
                               int T, int C, int H, int W) {
    if (t < 0 || h < 0 || w < 0 || t >= T || h >= H || w >= W) return (RK_T)0;
    return p[(((size_t)n * T + t) * C + c) * ((size_t)H * W) + (size_t)h * W + w];
}

/* ref3d:193-203 / :709-719 / :914-924 -- nested lerp, W innermost, then H, then T */
static inline RK_T RK_FN(trilerp)(const RK_T q[2][2][2], RK_T rT, RK_T rH, RK_T rW) {
    return (1 - rT) * ((1 - rH) * (q[0][0][0] * (1 - rW) + q[0][0][1] * rW) +
                       rH * (q[0][1][0] * (1 - rW) + q[0][1][1] * rW)) +
           rT * ((1 - rH) * (q[1][0][0] * (1 - rW) + q[1][0][1] * rW) +
                 rH * (q[1][1][0] * (1 - rW) + q[1][1][1] * rW));
}

/* ref3d:208-215 */
static inline RK_T RK_FN(interp2)(RK_T p11, RK_T p12, RK_T p21, RK_T p22, RK_T d1, RK_T d2) {
    return p11 * (1 - d1) * (1 - d2) + p12 * (1 - d1) * d2 + p21 * d1 * (1 - d2) + p22 * d1 * d2;
}

/* ------------------------------------------------------------------ */
/* K1: 3D forward -- ref3d:15-205                                      */
/* ------------------------------------------------------------------ */
void RK_FN(oracle_rk3d_forward)(const RK_T* x, const RK_T* shift, RK_T* y,
                                int N, int T, int C, int H, int W,
                                int To, int Ho, int Wo,
                                int sT, int sH, int sW, int pT, int pH, int pW,
                                int quantize) {
    const RK_T* shT = shift;           /* refcpp:243-244: rows T, H, W of the [3,C] buffer */
    const RK_T* shH = shift + C;
    const RK_T* shW = shift + 2 * C;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int to = 0; to < To; ++to)
            for (int c = 0; c < C; ++c) {
                const int flT = RK_FN(floor3d)(shT[c]);
                const int flH = RK_FN(floor3d)(shH[c]);
                const int flW = RK_FN(floor3d)(shW[c]);
                const RK_T rT = shT[c] - flT, rH = shH[c] - flH, rW = shW[c] - flW; /* ref3d:72-74 */
                RK_T* yp = y + (((size_t)n * To + to) * C + c) * ((size_t)Ho * Wo);
                for (int ho = 0; ho < Ho; ++ho)
                    for (int wo = 0; wo < Wo; ++wo) {
                        const int bT = to * sT - pT, bH = ho * sH - pH, bW = wo * sW - pW; /* :54-56 */
                        if (quantize) {                                       /* ref3d:76-93 */
                            const int qT = (rT < 0.5f) ? flT : flT + 1;
                            const int qH = (rH < 0.5f) ? flH : flH + 1;
                            const int qW = (rW < 0.5f) ? flW : flW + 1;
                            yp[ho * Wo + wo] = RK_FN(tap5)(x, n, bT + qT, c, bH + qH, bW + qW, T, C, H, W);
                            continue;
                        }
                        RK_T q[2][2][2];
                        for (int i = 0; i < 2; ++i)
                            for (int j = 0; j < 2; ++j)
                                for (int k = 0; k < 2; ++k)
                                    q[i][j][k] = RK_FN(tap5)(x, n, bT + flT + i, c, bH + flH + j,
                                                             bW + flW + k, T, C, H, W);
                        yp[ho * Wo + wo] = RK_FN(trilerp)(q, rT, rH, rW);
                    }
            }
}

/* ------------------------------------------------------------------ */
/* K2: 3D shift-gradient partials -- ref3d:218-452                     */
/* scratch is [3, C, Ho*Wo], must be zero on entry (refcpp:295); every */
/* output element adds its contribution (serial stand-in for atomicAdd;*/
/* per address the adds happen in increasing (n,to) order).            */
/* ------------------------------------------------------------------ */
void RK_FN(oracle_rk3d_backward_shift_partials)(const RK_T* x, const RK_T* shift, const RK_T* gy,
                                                RK_T* scratch,
                                                int N, int T, int C, int H, int W,
                                                int To, int Ho, int Wo,
                                                int sT, int sH, int sW, int pT, int pH, int pW) {
    const RK_T* shT = shift;
    const RK_T* shH = shift + C;
    const RK_T* shW = shift + 2 * C;
    const size_t HWo = (size_t)Ho * Wo;
    /* threads own (c, block of rows) of the scratch: every address still receives its adds in increasing (n, to)
     * order, so the sums are the serial ones; C * nblk work items keep all host cores busy (C alone is 64) while a
     * block of rows keeps each thread's reads of x / gy in runs of several rows */
    int nblk = 1;
#ifdef _OPENMP
    nblk = (omp_get_max_threads() + C - 1) / C;
#endif
    if (nblk < 1) nblk = 1;
    if (nblk > Ho) nblk = Ho;
    const int rows_per = (Ho + nblk - 1) / nblk;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
      for (int hb = 0; hb < nblk; ++hb) {
        const int flT = RK_FN(floor3d)(shT[c]);
        const int flH = RK_FN(floor3d)(shH[c]);
        const int flW = RK_FN(floor3d)(shW[c]);
        const RK_T rT = shT[c] - flT, rH = shH[c] - flH, rW = shW[c] - flW;   /* ref3d:283-285 */
        /* ref3d:288-298: an exactly-integer shift lowers the "small" index by one */
        const int zT = (rT == 0) ? 1 : 0, zH = (rH == 0) ? 1 : 0, zW = (rW == 0) ? 1 : 0;
        RK_T* accT = scratch + (size_t)c * HWo;
        RK_T* accH = scratch + ((size_t)C + c) * HWo;
        RK_T* accW = scratch + ((size_t)2 * C + c) * HWo;
        for (int n = 0; n < N; ++n)
            for (int to = 0; to < To; ++to) {
                const RK_T* gp = gy + (((size_t)n * To + to) * C + c) * HWo;
                for (int ho = hb * rows_per; ho < (hb + 1) * rows_per && ho < Ho; ++ho) {
                    for (int wo = 0; wo < Wo; ++wo) {
                        const int bT = to * sT - pT, bH = ho * sH - pH, bW = wo * sW - pW; /* :261-263 */
                        /* index 0 = "small" (possibly lowered), 1 = "large".  ref3d:359-431:
                         * each q***a is fetched at the lowered small index in every dimension
                         * whose remainder is exactly 0, else it equals the plain tap. */
                        const int tt[2] = {bT + flT - zT, bT + flT + 1};
                        const int hh[2] = {bH + flH - zH, bH + flH + 1};
                        const int ww[2] = {bW + flW - zW, bW + flW + 1};
                        RK_T q[2][2][2];
                        for (int i = 0; i < 2; ++i)
                            for (int j = 0; j < 2; ++j)
                                for (int k = 0; k < 2; ++k)
                                    q[i][j][k] = RK_FN(tap5)(x, n, tt[i], c, hh[j], ww[k], T, C, H, W);
                        /* ref3d:432-437 */
                        const RK_T Ts = RK_FN(interp2)(q[0][0][0], q[0][0][1], q[0][1][0], q[0][1][1], rH, rW);
                        const RK_T Tl = RK_FN(interp2)(q[1][0][0], q[1][0][1], q[1][1][0], q[1][1][1], rH, rW);
                        const RK_T Hs = RK_FN(interp2)(q[0][0][0], q[0][0][1], q[1][0][0], q[1][0][1], rT, rW);
                        const RK_T Hl = RK_FN(interp2)(q[0][1][0], q[0][1][1], q[1][1][0], q[1][1][1], rT, rW);
                        const RK_T Ws = RK_FN(interp2)(q[0][0][0], q[0][1][0], q[1][0][0], q[1][1][0], rT, rH);
                        const RK_T Wl = RK_FN(interp2)(q[0][0][1], q[0][1][1], q[1][0][1], q[1][1][1], rT, rH);
                        const RK_T up = gp[ho * Wo + wo];                         /* ref3d:443 */
                        accT[ho * Wo + wo] += (-Ts + Tl) * up;                    /* ref3d:439-450 */
                        accH[ho * Wo + wo] += (-Hs + Hl) * up;
                        accW[ho * Wo + wo] += (-Ws + Wl) * up;
                    }
                }
            }
      }
}

/* refcpp:344-345 (3D) and :140-143 (2D): shift_grad = scratch.view(D*C, HoWo) @ ones,
 * beta = 0 -- a plain row sum, any order; we sum left to right in T. */
void RK_FN(oracle_rowsum)(const RK_T* scratch, RK_T* out, int rows, int cols) {
    for (int r = 0; r < rows; ++r) {
        RK_T s = 0;
        for (int k = 0; k < cols; ++k) s += scratch[(size_t)r * cols + k];
        out[r] = s;
    }
}

/* K5: ref3d:932-960 */
void RK_FN(oracle_rk3d_normalize)(RK_T* gshift, int C, RK_T t_factor) {
    RK_T* gT = gshift;
    RK_T* gH = gshift + C;
    RK_T* gW = gshift + 2 * C;
    for (int c = 0; c < C; ++c) {
        RK_T a, b, d;
        if (t_factor < 0) { a = gT[c]; b = 0; d = 0; }
        else { a = gT[c] * t_factor; b = gH[c]; d = gW[c]; }
        const RK_T mag = (RK_T)sqrt((double)(a * a + b * b + d * d));
        if (mag > 0) { gT[c] = a / mag; gH[c] = b / mag; gW[c] = d / mag; }
    }
}

/* ------------------------------------------------------------------ */
/* K3 / K4: 3D input gradient -- ref3d:455-723 (generic) and :726-929  */
/* (stride 1 / pad 0).  K4 is K3 with the modulo tests and the pad     */
/* removed; `s1p0` selects it the way ref3d:1112-1113 does.            */
/* ------------------------------------------------------------------ */
static inline RK_T RK_FN(gtap)(const RK_T* gy, int n, int pt, int c, int ph, int pw,
                               int To, int C, int Ho, int Wo,
                               int sT, int sH, int sW, int s1p0) {
    if (!s1p0) {
        /* ref3d:586-589: C remainder semantics -- a negative non-multiple is != 0 */
        if (pt % sT != 0 || ph % sH != 0 || pw % sW != 0) return (RK_T)0;
        pt /= sT; ph /= sH; pw /= sW;
    }
    return RK_FN(tap5)(gy, n, pt, c, ph, pw, To, C, Ho, Wo);
}

void RK_FN(oracle_rk3d_backward_input)(const RK_T* shift, const RK_T* gy, RK_T* gx,
                                       int N, int T, int C, int H, int W,
                                       int To, int Ho, int Wo,
                                       int sT, int sH, int sW, int pT, int pH, int pW,
                                       int quantize) {
    const int s1p0 = (sT == 1 && sH == 1 && sW == 1 && pT == 0 && pH == 0 && pW == 0);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < T; ++t)
            for (int c = 0; c < C; ++c) {
                const RK_T nT = -shift[c], nH = -shift[C + c], nW = -shift[2 * C + c]; /* :505-507 */
                const int flT = RK_FN(floor3d)(nT), flH = RK_FN(floor3d)(nH), flW = RK_FN(floor3d)(nW);
                const RK_T rT = nT - flT, rH = nH - flH, rW = nW - flW;            /* ref3d:528-530 */
                RK_T* gp = gx + (((size_t)n * T + t) * C + c) * ((size_t)H * W);
                for (int h = 0; h < H; ++h)
                    for (int w = 0; w < W; ++w) {
                        const int oT = t + pT, oH = h + pH, oW = w + pW;           /* ref3d:498-500 */
                        RK_T val = 0;
                        if (quantize) {                                             /* ref3d:533-558 */
                            const int qT = (rT < 0.5f) ? flT : flT + 1;
                            const int qH = (rH < 0.5f) ? flH : flH + 1;
                            const int qW = (rW < 0.5f) ? flW : flW + 1;
                            val = RK_FN(gtap)(gy, n, oT + qT, c, oH + qH, oW + qW, To, C, Ho, Wo,
                                              sT, sH, sW, s1p0);
                        } else if (nT == 0 && nH == 0 && nW == 0) {                 /* ref3d:561-576 */
                            val = RK_FN(gtap)(gy, n, oT, c, oH, oW, To, C, Ho, Wo, sT, sH, sW, s1p0);
                        } else {                                                    /* ref3d:578-720 */
                            RK_T q[2][2][2];
                            for (int i = 0; i < 2; ++i)
                                for (int j = 0; j < 2; ++j)
                                    for (int k = 0; k < 2; ++k)
                                        q[i][j][k] = RK_FN(gtap)(gy, n, oT + flT + i, c, oH + flH + j,
                                                                 oW + flW + k, To, C, Ho, Wo,
                                                                 sT, sH, sW, s1p0);
                            val = RK_FN(trilerp)(q, rT, rH, rW);
                        }
                        gp[h * W + w] = val;                                        /* ref3d:721 */
                    }
            }
}

/* B2: the host-side composition of the 3D backward -- refcpp:256-379 */
void RK_FN(oracle_rk3d_backward)(const RK_T* x, const RK_T* shift, const RK_T* gy,
                                 RK_T* gx, RK_T* gshift, RK_T* scratch /* [3,C,Ho*Wo] */,
                                 int N, int T, int C, int H, int W,
                                 int To, int Ho, int Wo,
                                 int sT, int sH, int sW, int pT, int pH, int pW,
                                 int normalize_grad, RK_T t_factor, int quantize) {
    memset(scratch, 0, sizeof(RK_T) * 3 * (size_t)C * Ho * Wo);                   /* refcpp:295 */
    RK_FN(oracle_rk3d_backward_shift_partials)(x, shift, gy, scratch, N, T, C, H, W, To, Ho, Wo,
                                               sT, sH, sW, pT, pH, pW);           /* refcpp:324-338 */
    RK_FN(oracle_rowsum)(scratch, gshift, 3 * C, Ho * Wo);                        /* refcpp:344-345 */
    if (normalize_grad) RK_FN(oracle_rk3d_normalize)(gshift, C, t_factor);        /* refcpp:352-358 */
    RK_FN(oracle_rk3d_backward_input)(shift, gy, gx, N, T, C, H, W, To, Ho, Wo,
                                      sT, sH, sW, pT, pH, pW, quantize);          /* refcpp:363-376 */
}

/* ================================================================== */
/* 2D operator, layout [N,C,H,W], shift [2,C] = (H,W)                  */
/* ================================================================== */

static inline int RK_FN(floor_fast)(RK_T v) { int iv = (int)v; return iv - (v < iv); } /* ref2d:69-73 */
static inline int RK_FN(round_fast)(RK_T v) {                                           /* ref2d:76-82 */
    return (v < (RK_T)0.0f) ? (int)(v - (RK_T)0.5f) : (int)(v + (RK_T)0.5f);
}
/* ref2d:85-91: value only when in bounds, else the caller's default stays */
static inline int RK_FN(in4)(int h, int w, int H, int W) { return h >= 0 && w >= 0 && h < H && w < W; }
#define RK_AT4(p, n, c, h, w, C, H, W) (p)[(((size_t)(n) * (C) + (c)) * (H) + (h)) * (size_t)(W) + (w)]

/* ref2d:60-66 */
static inline RK_T RK_FN(interp2d)(RK_T px[2][2], RK_T rH, RK_T rW) {
    return px[0][0] * (1 - rH) * (1 - rW) + px[0][1] * (1 - rH) * rW + px[1][0] * rH * (1 - rW) +
           px[1][1] * rH * rW;
}

/* K6: ref2d:94-145.  NOTE the quantize branch writes nothing for an out-of-bounds
 * source (ref2d:116-121) -- y keeps whatever the caller put there (zeros from
 * allocate_output, rubiksnet/utils.py:26). */
void RK_FN(oracle_rk2d_forward)(const RK_T* x, const RK_T* shift, RK_T* y,
                                int N, int C, int H, int W, int Ho, int Wo,
                                int sH, int sW, int pH, int pW, int quantize) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const RK_T offH = shift[c], offW = shift[C + c];                      /* ref2d:113-114 */
            for (int ho = 0; ho < Ho; ++ho)
                for (int wo = 0; wo < Wo; ++wo) {
                    const int bH = ho * sH - pH, bW = wo * sW - pW;               /* ref2d:106-107 */
                    if (quantize) {
                        const int th = RK_FN(round_fast)(bH + offH), tw = RK_FN(round_fast)(bW + offW);
                        if (RK_FN(in4)(th, tw, H, W))
                            RK_AT4(y, n, c, ho, wo, C, Ho, Wo) = RK_AT4(x, n, c, th, tw, C, H, W);
                        continue;
                    }
                    const int iH = RK_FN(floor_fast)(offH), iW = RK_FN(floor_fast)(offW);
                    const RK_T rH = offH - iH, rW = offW - iW;                     /* ref2d:125-126 */
                    RK_T px[2][2] = {{0, 0}, {0, 0}};
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b)
                            if (RK_FN(in4)(bH + iH + a, bW + iW + b, H, W))
                                px[a][b] = RK_AT4(x, n, c, bH + iH + a, bW + iW + b, C, H, W);
                    RK_AT4(y, n, c, ho, wo, C, Ho, Wo) = RK_FN(interp2d)(px, rH, rW);
                }
        }
}

/* K7: ref2d:147-266.  scratch [2, C, Ho, Wo], zero on entry. */
void RK_FN(oracle_rk2d_backward_shift_partials)(const RK_T* gy, const RK_T* x, const RK_T* shift,
                                                RK_T* scratch,
                                                int N, int C, int H, int W, int Ho, int Wo,
                                                int sH, int sW, int pH, int pW) {
    const size_t HWo = (size_t)Ho * Wo;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const RK_T offH = shift[c], offW = shift[C + c];
        const int iH = RK_FN(floor_fast)(offH), iW = RK_FN(floor_fast)(offW);     /* ref2d:173-174 */
        RK_T rH = offH - iH, rW = offW - iW;                                       /* ref2d:179-180 */
        const RK_T tol = (RK_T)1e-7f;                                              /* ref2d:189 */
        int hint = 0, wint = 0;
        if (tol > rH && rH > -tol) { hint = 1; rH = 0; }                           /* ref2d:193-200 */
        if (tol > rW && rW > -tol) { wint = 1; rW = 0; }
        for (int n = 0; n < N; ++n)
            for (int ho = 0; ho < Ho; ++ho)
                for (int wo = 0; wo < Wo; ++wo) {
                    const int h0 = ho * sH - pH + iH, w0 = wo * sW - pW + iW;      /* ref2d:160-177 */
                    RK_T px[2][2] = {{0, 0}, {0, 0}};
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b)
                            if (RK_FN(in4)(h0 + a, w0 + b, H, W))
                                px[a][b] = RK_AT4(x, n, c, h0 + a, w0 + b, C, H, W);
                    /* ref2d:215-221 */
                    RK_T dH = (1 - rW) * (px[1][0] - px[0][0]) + rW * (px[1][1] - px[0][1]);
                    RK_T dW = (1 - rH) * (px[0][1] - px[0][0]) + rH * (px[1][1] - px[1][0]);
                    if (hint || wint) {                                            /* ref2d:224-253 */
                        RK_T p3[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                        for (int a = 0; a < 3; ++a)
                            for (int b = 0; b < 3; ++b) {
                                if ((a == 0 && b == 0) || (a == 1 && b == 1)) continue;
                                if (RK_FN(in4)(h0 + a - 1, w0 + b - 1, H, W))
                                    p3[a][b] = RK_AT4(x, n, c, h0 + a - 1, w0 + b - 1, C, H, W);
                            }
                        if (hint)
                            dH = (RK_T)0.5f * ((1 - rW) * (p3[2][1] - p3[0][1]) + rW * (p3[2][2] - p3[0][2]));
                        if (wint)
                            dW = (RK_T)0.5f * ((1 - rH) * (p3[1][2] - p3[1][0]) + rH * (p3[2][2] - p3[2][0]));
                    }
                    const RK_T og = RK_AT4(gy, n, c, ho, wo, C, Ho, Wo);           /* ref2d:256-264 */
                    scratch[(size_t)c * HWo + (size_t)ho * Wo + wo] += dH * og;
                    scratch[((size_t)C + c) * HWo + (size_t)ho * Wo + wo] += dW * og;
                }
    }
}

/* K8: ref2d:269-379.  In-kernel stride/pad are uint32 (ref2d:274-275); with a
 * negative int operand the mixed %,/ wrap to huge values that always fail either
 * the divisibility or the bounds test, i.e. negatives are skipped -- restated here
 * through the same unsigned arithmetic.  The quantize branch writes nothing when
 * it skips (ref2d:294-309). */
static inline int RK_FN(unmap2d)(int* v, unsigned s, int lim) {
    unsigned u = (unsigned)*v;
    if (u % s != 0) return 0;
    int q = (int)(u / s);
    if (q < 0 || (unsigned)q >= (unsigned)lim) return 0;
    *v = q;
    return 1;
}

void RK_FN(oracle_rk2d_backward_input)(const RK_T* gy, const RK_T* shift, RK_T* gx,
                                       int N, int C, int H, int W, int Ho, int Wo,
                                       int sH, int sW, int pH, int pW, int quantize) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const RK_T nH = -shift[c], nW = -shift[C + c];                         /* ref2d:291-292 */
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    const int oH = h + pH, oW = w + pW;                            /* ref2d:285-286 */
                    if (quantize) {
                        int th = RK_FN(round_fast)(oH + nH), tw = RK_FN(round_fast)(oW + nW);
                        /* ref2d:298: both divisibility tests come before either division */
                        if ((unsigned)th % (unsigned)sH == 0 && (unsigned)tw % (unsigned)sW == 0 &&
                            RK_FN(unmap2d)(&th, (unsigned)sH, Ho) && RK_FN(unmap2d)(&tw, (unsigned)sW, Wo))
                            RK_AT4(gx, n, c, h, w, C, H, W) = RK_AT4(gy, n, c, th, tw, C, Ho, Wo);
                        continue;
                    }
                    RK_T val = 0;
                    const int flH = RK_FN(floor_fast)(nH), flW = RK_FN(floor_fast)(nW); /* :316-319 */
                    if (nW == 0 && nH == 0) {                                      /* ref2d:322-339 */
                        int th = oH, tw = oW;
                        if ((unsigned)tw % (unsigned)sW == 0 && (unsigned)th % (unsigned)sH == 0 &&
                            RK_FN(unmap2d)(&th, (unsigned)sH, Ho) && RK_FN(unmap2d)(&tw, (unsigned)sW, Wo))
                            val = RK_AT4(gy, n, c, th, tw, C, Ho, Wo);
                    } else {                                                       /* ref2d:341-376 */
                        RK_T px[2][2] = {{0, 0}, {0, 0}};
                        for (int a = 0; a < 2; ++a)
                            for (int b = 0; b < 2; ++b) {
                                int th = oH + flH + a, tw = oW + flW + b;
                                if ((unsigned)th % (unsigned)sH == 0 && (unsigned)tw % (unsigned)sW == 0 &&
                                    RK_FN(unmap2d)(&th, (unsigned)sH, Ho) &&
                                    RK_FN(unmap2d)(&tw, (unsigned)sW, Wo))
                                    px[a][b] = RK_AT4(gy, n, c, th, tw, C, Ho, Wo);
                            }
                        const RK_T rH = nH - flH, rW = nW - flW;                    /* ref2d:373-374 */
                        val = RK_FN(interp2d)(px, rH, rW);
                    }
                    RK_AT4(gx, n, c, h, w, C, H, W) = val;                         /* ref2d:377 */
                }
        }
}

/* K9: ref2d:381-397 */
void RK_FN(oracle_rk2d_normalize)(RK_T* gshift, int C) {
    for (int c = 0; c < C; ++c) {
        const RK_T a = gshift[c], b = gshift[C + c];
        const RK_T mag = (RK_T)sqrt((double)(a * a + b * b));
        if (mag > 0) { gshift[c] = a / mag; gshift[C + c] = b / mag; }
    }
}

/* B4: refcpp:94-155.  gx / gshift are caller-zeroed (refcpp:106-107). */
void RK_FN(oracle_rk2d_backward)(const RK_T* gy, const RK_T* x, const RK_T* shift,
                                 RK_T* gx, RK_T* gshift, RK_T* scratch /* [2,C,Ho,Wo] */,
                                 int N, int C, int H, int W, int Ho, int Wo,
                                 int sH, int sW, int pH, int pW,
                                 int normalize_grad, int enable_shift_grad, int quantize) {
    if (enable_shift_grad) {                                                        /* refcpp:126-149 */
        memset(scratch, 0, sizeof(RK_T) * 2 * (size_t)C * Ho * Wo);
        RK_FN(oracle_rk2d_backward_shift_partials)(gy, x, shift, scratch, N, C, H, W, Ho, Wo,
                                                   sH, sW, pH, pW);
        RK_FN(oracle_rowsum)(scratch, gshift, 2 * C, Ho * Wo);
        if (normalize_grad) RK_FN(oracle_rk2d_normalize)(gshift, C);
    }
    RK_FN(oracle_rk2d_backward_input)(gy, shift, gx, N, C, H, W, Ho, Wo, sH, sW, pH, pW, quantize);
}

#undef RK_AT4
