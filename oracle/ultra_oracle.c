/*
 * oracle/ultra_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See ultra_oracle.h.  CPU restatement of the secup/ProjectUltra receive hot
 * path (references are into /root/reference).  Parity is PINNED against the
 * compiled reference and its known-answer tests (tests/test_oracle_*.py).
 *
 * Arithmetic rules that decide bit-exactness:
 *   - the reference is built without -march / fast-math, so no FMA contraction:
 *     this file is compiled with -ffp-contract=off;
 *   - M_PI is double: every `2.0f * M_PI * ...` is evaluated in double and
 *     narrowed on assignment — restated literally below;
 *   - std::complex<float> `*`  = (ac-bd, ad+bc) in float (finite operands);
 *     std::complex<float> `/`  = libgcc __divsc3, which in the libgcc_s.so.1
 *     of this image (GCC 12.3) evaluates in double:
 *       x=(a*c+b*d)/(c*c+d*d), y=(b*c-a*d)/(c*c+d*d), each narrowed once
 *     (checked 1e7/1e7 against g++ here, tests/test_oracle_vs_ref.py);
 *   - complex/float and float*complex are component-wise;
 *   - std::abs(complex)=hypotf, std::arg=atan2f, std::norm=re*re+im*im,
 *     std::exp(Complex(0,t)) = cexpf = (cosf t, sinf t) via sincosf;
 *   - libm calls (cosf, sinf, atan2f, hypotf, powf) are the host glibc's, the
 *     same functions the reference binary calls.
 */
#define _GNU_SOURCE
#include "ultra_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ====================================================================== */
/* complex helpers                                                        */
/* ====================================================================== */
typedef struct cf { float re, im; } cf;

static inline cf c_make(float re, float im) { cf r = {re, im}; return r; }
static inline cf c_add(cf a, cf b) { return c_make(a.re + b.re, a.im + b.im); }
static inline cf c_sub(cf a, cf b) { return c_make(a.re - b.re, a.im - b.im); }
static inline cf c_conj(cf a) { return c_make(a.re, -a.im); }
/* std::complex<float> operator* (GCC inline expansion, finite operands) */
static inline cf c_mul(cf x, cf y) {
    float ac = x.re * y.re, bd = x.im * y.im, ad = x.re * y.im, bc = x.im * y.re;
    return c_make(ac - bd, ad + bc);
}
/* libgcc __divsc3 as shipped in this image (double evaluation) */
static inline cf c_div(cf x, cf y) {
    double a = x.re, b = x.im, c = y.re, d = y.im;
    double denom = (c * c) + (d * d);
    double xr = ((a * c) + (b * d)) / denom;
    double yi = ((b * c) - (a * d)) / denom;
    return c_make((float)xr, (float)yi);
}
static inline cf c_scale(cf a, float s) { return c_make(a.re * s, a.im * s); }   /* complex * float */
static inline cf c_divf(cf a, float s) { return c_make(a.re / s, a.im / s); }    /* complex / float */
static inline float c_norm(cf a) { return a.re * a.re + a.im * a.im; }            /* std::norm */
static inline float c_abs(cf a) { return hypotf(a.re, a.im); }                    /* std::abs  */
static inline float c_arg(cf a) { return atan2f(a.im, a.re); }                    /* std::arg  */
/* std::exp(Complex(0, t)) → glibc cexpf: expf(0)=1, sincosf(t) */
static inline cf c_expj(float t) { return c_make(cosf(t), sinf(t)); }
static inline float f_min(float a, float b) { return (b < a) ? b : a; }           /* std::min(a,b) */
static inline float f_max(float a, float b) { return (a < b) ? b : a; }           /* std::max(a,b) */

/* ====================================================================== */
/* mt19937 (std::mt19937; pinned by reference tests/test_rng.cpp:24-39)   */
/* ====================================================================== */
void uo_mt_seed(uo_mt19937* r, uint32_t seed) {
    r->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    r->idx = 624;
}
uint32_t uo_mt_next(uo_mt19937* r) {
    if (r->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (r->mt[i] & 0x80000000u) | (r->mt[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = r->mt[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            r->mt[i] = v;
        }
        r->idx = 0;
    }
    uint32_t y = r->mt[r->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* ====================================================================== */
/* geometry                                                               */
/* ====================================================================== */
/* getBitsPerSymbol, include/ultra/types.hpp:42-56 */
static uint32_t bits_per_symbol(uint32_t mod) {
    switch (mod) {
        case ULTRA_MOD_DBPSK: case ULTRA_MOD_BPSK: return 1;
        case ULTRA_MOD_DQPSK: case ULTRA_MOD_QPSK: return 2;
        case ULTRA_MOD_D8PSK: case ULTRA_MOD_QAM8: return 3;
        case ULTRA_MOD_QAM16: return 4;
        case ULTRA_MOD_QAM32: return 5;
        case ULTRA_MOD_QAM64: return 6;
        case ULTRA_MOD_QAM256: return 8;
        default: return 1;
    }
}
/* ModemConfig::getCyclicPrefix, include/ultra/types.hpp:197-208 */
static uint32_t cyclic_prefix(const ultra_hip_config* c) {
    uint32_t base;
    switch (c->cp_mode) {
        case ULTRA_CP_SHORT: base = 32; break;
        case ULTRA_CP_MEDIUM: base = 48; break;
        case ULTRA_CP_LONG: base = 64; break;
        default: base = 48;
    }
    return base * (c->fft_size / 512);
}
/* getCodeParams, src/fec/ldpc_decoder.cpp:22-36 */
static void code_params(uint32_t rate, int* k, int* m) {
    switch (rate) {
        case ULTRA_RATE_R1_4: *k = 162; *m = 486; break;
        case ULTRA_RATE_R1_2: *k = 324; *m = 324; break;
        case ULTRA_RATE_R2_3: *k = 432; *m = 216; break;
        case ULTRA_RATE_R3_4: *k = 486; *m = 162; break;
        case ULTRA_RATE_R5_6: *k = 540; *m = 108; break;
        default: *k = 324; *m = 324; break;
    }
}

/* ====================================================================== */
/* LDPC                                                                   */
/* ====================================================================== */
#define LDPC_N 648
#define LDPC_MAX_EDGES 4096

typedef struct ldpc_code {
    uint32_t rate;
    int k, m, edges;
    uint32_t row_ptr[487];
    uint16_t col[LDPC_MAX_EDGES];
} ldpc_code;

/* LDPCDecoder::Impl::buildMatrix, src/fec/ldpc_decoder.cpp:64-137
 * (== LDPCEncoder::Impl::buildMatrix, src/fec/ldpc_encoder.cpp:70-129) */
static void ldpc_build(ldpc_code* code, uint32_t rate) {
    int k, m;
    code_params(rate, &k, &m);
    code->rate = rate; code->k = k; code->m = m;
    uo_mt19937 rng;
    uo_mt_seed(&rng, 0x12345678u + (uint32_t)(int)rate);

    /* rows as growable lists (max row degree = 6 info + 1 parity) */
    static const int MAXDEG = 8;
    int* rows = (int*)malloc(sizeof(int) * (size_t)m * MAXDEG);
    int* deg = (int*)calloc((size_t)m, sizeof(int));
    int* check_degrees = (int*)calloc((size_t)m, sizeof(int));
    int* avail = (int*)malloc(sizeof(int) * (size_t)m);

    int target_check_degree = 4;
    int target_var_degree = (target_check_degree * m) / k;
    if (target_var_degree < 3) target_var_degree = 3;
    if (target_var_degree > m / 2) target_var_degree = m / 2;
    int max_check_degree = target_check_degree + 2;

    for (int j = 0; j < k; ++j) {
        int na = 0;
        for (int i = 0; i < m; ++i)
            if (check_degrees[i] < max_check_degree) avail[na++] = i;
        /* Fisher-Yates from the top with rng() % i (decoder.cpp:92-95) */
        for (size_t i = (size_t)na; i > 1; --i) {
            size_t jj = uo_mt_next(&rng) % i;
            int t = avail[i - 1]; avail[i - 1] = avail[jj]; avail[jj] = t;
        }
        int connections = target_var_degree < na ? target_var_degree : na;
        for (int d = 0; d < connections; ++d) {
            int check = avail[d];
            rows[check * MAXDEG + deg[check]++] = j;
            check_degrees[check]++;
        }
    }
    /* every check gets at least one info bit (decoder.cpp:106-113) */
    for (int i = 0; i < m; ++i) {
        if (deg[i] == 0) {
            int j = (int)(uo_mt_next(&rng) % (uint32_t)k);
            rows[i * MAXDEG + deg[i]++] = j;
        }
    }
    /* identity part (decoder.cpp:115-120) */
    for (int i = 0; i < m; ++i) rows[i * MAXDEG + deg[i]++] = k + i;

    int e = 0;
    for (int i = 0; i < m; ++i) {
        code->row_ptr[i] = (uint32_t)e;
        for (int d = 0; d < deg[i]; ++d) code->col[e++] = (uint16_t)rows[i * MAXDEG + d];
    }
    code->row_ptr[m] = (uint32_t)e;
    code->edges = e;
    free(rows); free(deg); free(check_degrees); free(avail);
}

static pthread_mutex_t g_code_lock = PTHREAD_MUTEX_INITIALIZER;
static ldpc_code g_codes[8];
static int g_code_ready[8];
static const ldpc_code* ldpc_get(uint32_t rate) {
    uint32_t slot = rate < 8 ? rate : 7;
    pthread_mutex_lock(&g_code_lock);
    if (!g_code_ready[slot]) { ldpc_build(&g_codes[slot], rate); g_code_ready[slot] = 1; }
    pthread_mutex_unlock(&g_code_lock);
    return &g_codes[slot];
}

int uo_ldpc_graph(uint32_t rate, uint32_t* row_ptr, uint32_t* col_idx, uint32_t* k, uint32_t* m) {
    const ldpc_code* c = ldpc_get(rate);
    if (row_ptr) for (int i = 0; i <= c->m; ++i) row_ptr[i] = c->row_ptr[i];
    if (col_idx) for (int e = 0; e < c->edges; ++e) col_idx[e] = c->col[e];
    if (k) *k = (uint32_t)c->k;
    if (m) *m = (uint32_t)c->m;
    return c->edges;
}

/* LDPCEncoder::encode, src/fec/ldpc_encoder.cpp:193-257: bit-level blocks of
 * k info bits, parity = H_data·info, each codeword packed to bytes MSB-first. */
int uo_ldpc_encode(uint32_t rate, const uint8_t* data, uint32_t n, uint8_t* out, uint32_t cap) {
    const ldpc_code* c = ldpc_get(rate);
    int k = c->k, m = c->m, nn = k + m;
    size_t total_bits = (size_t)n * 8, bit_offset = 0;
    uint32_t o = 0;
    uint8_t cw[LDPC_N];
    while (bit_offset < total_bits) {
        for (int j = 0; j < k; ++j) {
            size_t b = bit_offset + (size_t)j;
            cw[j] = (b < total_bits) ? (uint8_t)((data[b / 8] >> (7 - (b % 8))) & 1) : 0;
        }
        for (int i = 0; i < m; ++i) {
            uint8_t s = 0;
            /* H_data rows = row minus the trailing identity edge */
            for (uint32_t e = c->row_ptr[i]; e + 1 < c->row_ptr[i + 1]; ++e) s ^= cw[c->col[e]];
            cw[k + i] = s;
        }
        uint8_t byte = 0; int bc = 0;
        for (int j = 0; j < nn; ++j) {
            byte = (uint8_t)((byte << 1) | cw[j]);
            if (++bc == 8) { if (o >= cap) return -1; out[o++] = byte; byte = 0; bc = 0; }
        }
        if (bc > 0) { if (o >= cap) return -1; out[o++] = (uint8_t)(byte << (8 - bc)); }
        bit_offset += (size_t)k;
    }
    return (int)o;
}

/* Impl::decodeBP core, src/fec/ldpc_decoder.cpp:153-236 (and the inline copy
 * in decodeSoft :301-375).  llr_total is left holding the final totals. */
static void ldpc_decode_block(const ldpc_code* c, int max_iters, const float* llr, int n_llr,
                              float* llr_total, int* success, int* iters) {
    int m = c->m, n = c->k + c->m;
    float llr_in[LDPC_N];
    float v2c[LDPC_MAX_EDGES], c2v[LDPC_MAX_EDGES];
    for (int j = 0; j < n; ++j) { llr_in[j] = 0; llr_total[j] = 0; }
    for (int j = 0; j < n && j < n_llr; ++j) { llr_in[j] = llr[j]; llr_total[j] = llr[j]; }
    for (int e = 0; e < c->edges; ++e) { v2c[e] = llr_in[c->col[e]]; c2v[e] = 0; }

    int ok = 0, it;
    for (it = 0; it < max_iters; ++it) {
        /* check-to-variable, brute-force "all others" (:181-202) */
        for (int i = 0; i < m; ++i) {
            uint32_t e0 = c->row_ptr[i], e1 = c->row_ptr[i + 1];
            for (uint32_t e = e0; e < e1; ++e) {
                float sign = 1.0f, min_abs = 3.402823466e+38f;
                for (uint32_t e2 = e0; e2 < e1; ++e2) {
                    if (e2 != e) {
                        float msg = v2c[e2];
                        if (msg < 0) sign = -sign;
                        float a = fabsf(msg);
                        if (a < min_abs) min_abs = a;
                    }
                }
                c2v[e] = sign * min_abs * 0.75f;
            }
        }
        /* totals, row-major accumulation (:206-213) */
        for (int j = 0; j < n; ++j) llr_total[j] = llr_in[j];
        for (int e = 0; e < c->edges; ++e) llr_total[c->col[e]] += c2v[e];
        /* variable-to-check with clamp (:216-224) */
        for (int e = 0; e < c->edges; ++e) {
            float v = llr_total[c->col[e]] - c2v[e];
            v2c[e] = f_max(-50.0f, f_min(50.0f, v));
        }
        /* hard decision + parity (:227-235, checkParity :139-151) */
        int pass = 1;
        for (int i = 0; i < m && pass; ++i) {
            uint8_t s = 0;
            for (uint32_t e = c->row_ptr[i]; e < c->row_ptr[i + 1]; ++e)
                s ^= (uint8_t)((llr_total[c->col[e]] < 0) ? 1 : 0);
            if (s) pass = 0;
        }
        if (pass) { ok = 1; break; }
    }
    *success = ok;
    *iters = it; /* index of the successful iteration, or max_iters */
}

static int pack_bits(const uint8_t* bits, int nbits, uint8_t* out, uint32_t cap) {
    uint32_t o = 0; uint8_t byte = 0; int bc = 0;
    for (int j = 0; j < nbits; ++j) {
        byte = (uint8_t)((byte << 1) | bits[j]);
        if (++bc == 8) { if (o >= cap) return -1; out[o++] = byte; byte = 0; bc = 0; }
    }
    if (bc > 0) { if (o >= cap) return -1; out[o++] = (uint8_t)(byte << (8 - bc)); }
    return (int)o;
}

/* LDPCDecoder::decodeSoft, src/fec/ldpc_decoder.cpp:283-428 */
int uo_ldpc_decode_soft(uint32_t rate, int max_iters, const float* llr, uint32_t n_llr,
                        uint8_t* out, uint32_t cap, int* success, int* iters) {
    const ldpc_code* c = ldpc_get(rate);
    int n = c->k + c->m, k = c->k;
    float total[LDPC_N];
    if (n_llr == 0) { *success = 0; return 0; }
    if (n_llr <= (uint32_t)n) {
        ldpc_decode_block(c, max_iters, llr, (int)n_llr, total, success, iters);
        uint8_t bits[LDPC_N];
        for (int j = 0; j < k; ++j) bits[j] = (total[j] < 0) ? 1 : 0;
        return pack_bits(bits, k, out, cap);
    }
    size_t nblocks = (n_llr + (uint32_t)n - 1) / (uint32_t)n;
    uint8_t* all = (uint8_t*)malloc(nblocks * (size_t)k);
    size_t nb = 0, off = 0;
    int all_ok = 1, ok;
    while (off + (size_t)n <= n_llr) {
        ldpc_decode_block(c, max_iters, llr + off, n, total, &ok, iters);
        if (!ok) all_ok = 0;
        for (int j = 0; j < k; ++j) all[nb++] = (total[j] < 0) ? 1 : 0;
        off += (size_t)n;
    }
    *success = all_ok;
    if (off < n_llr) { /* zero-padded tail block via decodeBP (:396-407): sets last_success itself */
        ldpc_decode_block(c, max_iters, llr + off, (int)(n_llr - off), total, &ok, iters);
        *success = ok;
        for (int j = 0; j < k; ++j) all[nb++] = (total[j] < 0) ? 1 : 0;
    }
    int r = pack_bits(all, (int)nb, out, cap);
    free(all);
    return r;
}

int uo_ldpc_decode_batch(uint32_t rate, int max_iters, const float* llr, uint32_t n_cw,
                         uint8_t* out, uint32_t bytes_per_cw, int32_t* iters, uint8_t* ok,
                         float* llr_total_out) {
    const ldpc_code* c = ldpc_get(rate);
    int k = c->k;
    if ((uint32_t)((k + 7) / 8) != bytes_per_cw) return -1;
    for (uint32_t w = 0; w < n_cw; ++w) {
        float total[LDPC_N];
        int s, it;
        ldpc_decode_block(c, max_iters, llr + (size_t)LDPC_N * w, LDPC_N, total, &s, &it);
        uint8_t bits[LDPC_N];
        for (int j = 0; j < k; ++j) bits[j] = (total[j] < 0) ? 1 : 0;
        pack_bits(bits, k, out + (size_t)bytes_per_cw * w, bytes_per_cw);
        iters[w] = it; ok[w] = (uint8_t)s;
        if (llr_total_out) memcpy(llr_total_out + (size_t)LDPC_N * w, total, sizeof(total));
    }
    return 0;
}

/* The same over n_threads worker threads (disjoint codeword ranges): the timed CPU leg of the LDPC-only bench. */
typedef struct dec_job { uint32_t rate; int max_iters; const float* llr; uint32_t n0, n1; uint8_t* out; uint32_t bpc;
                         int32_t* iters; uint8_t* ok; int rc; } dec_job;
static void* dec_worker(void* arg) {
    dec_job* j = (dec_job*)arg;
    j->rc = uo_ldpc_decode_batch(j->rate, j->max_iters, j->llr + (size_t)LDPC_N * j->n0, j->n1 - j->n0,
                                 j->out + (size_t)j->bpc * j->n0, j->bpc, j->iters + j->n0, j->ok + j->n0, NULL);
    return NULL;
}
int uo_ldpc_decode_batch_mt(uint32_t rate, int max_iters, const float* llr, uint32_t n_cw, int n_threads,
                            uint8_t* out, uint32_t bytes_per_cw, int32_t* iters, uint8_t* ok) {
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > n_cw) n_threads = n_cw ? (int)n_cw : 1;
    (void)ldpc_get(rate);                              /* build the code before the threads start */
    dec_job* jobs = (dec_job*)calloc((size_t)n_threads, sizeof(dec_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; ++t) {
        dec_job* j = &jobs[t];
        j->rate = rate; j->max_iters = max_iters; j->llr = llr; j->out = out; j->bpc = bytes_per_cw; j->iters = iters; j->ok = ok;
        j->n0 = (uint32_t)((uint64_t)n_cw * (uint64_t)t / (uint64_t)n_threads);
        j->n1 = (uint32_t)((uint64_t)n_cw * (uint64_t)(t + 1) / (uint64_t)n_threads);
        if (n_threads == 1) dec_worker(j); else pthread_create(&th[t], NULL, dec_worker, j);
    }
    int rc = 0;
    for (int t = 0; t < n_threads; ++t) { if (n_threads > 1) pthread_join(th[t], NULL); if (jobs[t].rc) rc = jobs[t].rc; }
    free(jobs); free(th);
    return rc;
}

/* Interleaver(rows, cols)::deinterleave(soft), src/fec/ldpc_decoder.cpp:454-466,530-540 */
int uo_interleaver_deinterleave(uint32_t rows, uint32_t cols, const float* in, uint32_t n, float* out) {
    size_t np = (size_t)rows * cols;
    for (uint32_t i = 0; i < n; ++i) out[i] = 0.0f;
    for (size_t i = 0; i < n && i < np; ++i) {
        size_t row = i / cols, col = i % cols;
        size_t p = col * rows + row;
        out[i] = in[p];
    }
    return (int)n;
}

/* ChannelInterleaver ctor, src/fec/ldpc_decoder.cpp:547-620:
 * perm[i] = (i*step) % total, inv[perm[i]] = i; deinterleave: out[inv[i]] = in[i] */
int uo_channel_interleaver_perm(uint32_t bits_per_symbol, uint32_t total, uint32_t* perm, uint32_t* inv) {
    size_t n = bits_per_symbol, tot = total;
    size_t target = n * 3;
    if (target >= tot) target = tot / 2;
    size_t step = 0; int found = 0;
    for (size_t s = target; s < tot && !found; s++) {
        size_t a = s, b = tot;
        while (b != 0) { size_t t = b; b = a % b; a = t; }
        if (a == 1) { step = s; found = 1; }
    }
    for (size_t s = n + 1; s < tot && !found; s++) {
        size_t a = s, b = tot;
        while (b != 0) { size_t t = b; b = a % b; a = t; }
        if (a == 1) { step = s; found = 1; }
    }
    if (!found) step = n + 1;
    for (size_t i = 0; i < tot; ++i) {
        size_t dest = (i * step) % tot;
        perm[i] = (uint32_t)dest;
        inv[dest] = (uint32_t)i;
    }
    return (int)step;
}

/* ====================================================================== */
/* FFT (built-in radix-2 path, src/dsp/fft.cpp:75-82,89-121) and NCO      */
/* ====================================================================== */
typedef struct fft_plan { uint32_t n; cf* tw; } fft_plan;

static void fft_init(fft_plan* p, uint32_t n) {
    p->n = n;
    p->tw = (cf*)malloc(sizeof(cf) * (n / 2 ? n / 2 : 1));
    for (size_t k = 0; k < n / 2; ++k) {
        /* float angle = -2.0f * M_PI * k / size;  (double expr, narrowed) */
        float angle = (float)(((double)-2.0f * M_PI * (double)k) / (double)n);
        p->tw[k] = c_make(cosf(angle), sinf(angle));
    }
}
static void fft_free(fft_plan* p) { free(p->tw); p->tw = NULL; }

static void fft_exec(const fft_plan* p, cf* data, int inverse) {
    size_t size = p->n;
    size_t j = 0;
    for (size_t i = 0; i + 1 < size; ++i) {
        if (i < j) { cf t = data[i]; data[i] = data[j]; data[j] = t; }
        size_t k = size / 2;
        while (k <= j) { j -= k; k /= 2; }
        j += k;
    }
    for (size_t len = 2; len <= size; len *= 2) {
        size_t half = len / 2, step = size / len;
        for (size_t i = 0; i < size; i += len) {
            for (size_t k = 0; k < half; ++k) {
                cf w = p->tw[k * step];
                if (inverse) w = c_conj(w);
                cf t = c_mul(w, data[i + k + half]);
                data[i + k + half] = c_sub(data[i + k], t);
                data[i + k] = c_add(data[i + k], t);
            }
        }
    }
    if (inverse) {
        float scale = 1.0f / (float)size;
        for (size_t i = 0; i < size; ++i) data[i] = c_scale(data[i], scale);
    }
}

int uo_fft_forward(uint32_t n, const float* in_ri, float* out_ri) {
    fft_plan p; fft_init(&p, n);
    memcpy(out_ri, in_ri, sizeof(float) * 2 * n);
    fft_exec(&p, (cf*)out_ri, 0);
    fft_free(&p);
    return 0;
}
int uo_fft_inverse(uint32_t n, const float* in_ri, float* out_ri) {
    fft_plan p; fft_init(&p, n);
    memcpy(out_ri, in_ri, sizeof(float) * 2 * n);
    fft_exec(&p, (cf*)out_ri, 1);
    fft_free(&p);
    return 0;
}

/* NCO, src/dsp/filters.cpp:228-238 */
typedef struct nco { float phase, inc; } nco;
static void nco_init(nco* o, float frequency, float sample_rate) {
    o->phase = 0;
    o->inc = (float)(((double)2.0f * M_PI * (double)frequency) / (double)sample_rate);
}
static inline cf nco_next(nco* o) {
    cf out = c_make(cosf(o->phase), sinf(o->phase));
    o->phase += o->inc;
    if ((double)o->phase > (double)2.0f * M_PI) o->phase = (float)((double)o->phase - (double)2.0f * M_PI);
    if (o->phase < 0) o->phase = (float)((double)o->phase + (double)2.0f * M_PI);
    return out;
}
int uo_nco(float freq, float fs, uint32_t n, float* out_ri) {
    nco o; nco_init(&o, freq, fs);
    for (uint32_t i = 0; i < n; ++i) { cf c = nco_next(&o); out_ri[2 * i] = c.re; out_ri[2 * i + 1] = c.im; }
    return 0;
}

/* ====================================================================== */
/* Carrier layout shared by modulator and demodulator                      */
/* ====================================================================== */
#define MAX_CARRIERS 128
#define MAX_FFT 1024

typedef struct carriers {
    int n_data, n_pilot;
    int data_idx[MAX_CARRIERS];
    int pilot_idx[MAX_CARRIERS];
    cf pilot_seq[MAX_CARRIERS];
    int n_sync;
    cf sync_seq[MAX_CARRIERS];
} carriers;

/* setupCarriers + generateSequences (sync + pilot sequences),
 * src/ofdm/demodulator.cpp:46-85 == src/ofdm/modulator.cpp:141-200 */
static int carriers_init(carriers* cr, const ultra_hip_config* c) {
    if (c->num_carriers == 0 || c->num_carriers > MAX_CARRIERS || c->fft_size > MAX_FFT) return -1;
    if (c->use_pilots && c->pilot_spacing == 0) return -1;
    int neg_limit = (int)(c->num_carriers / 2);
    int pos_limit = (int)((c->num_carriers + 1) / 2);
    cr->n_data = cr->n_pilot = 0;
    int pilot_count = 0;
    for (int i = -neg_limit; i <= pos_limit; ++i) {
        if (i == 0) continue;
        int fft_idx = (int)(((unsigned)i + c->fft_size) % c->fft_size); /* (i + fft_size) % fft_size in unsigned */
        if (!c->use_pilots) cr->data_idx[cr->n_data++] = fft_idx;
        else if (pilot_count % (int)c->pilot_spacing == 0) cr->pilot_idx[cr->n_pilot++] = fft_idx;
        else cr->data_idx[cr->n_data++] = fft_idx;
        ++pilot_count;
    }
    size_t N = c->num_carriers, u = 1;
    cr->n_sync = (int)N;
    for (size_t n = 0; n < N; ++n) {
        /* float phase = -M_PI * u * n * (n + 1) / N; */
        float phase = (float)((((-M_PI * (double)u) * (double)n) * (double)(n + 1)) / (double)N);
        cr->sync_seq[n] = c_make(cosf(phase), sinf(phase));
    }
    uo_mt19937 rng; uo_mt_seed(&rng, 0x50494C54u);
    for (int i = 0; i < cr->n_pilot; ++i)
        cr->pilot_seq[i] = (uo_mt_next(&rng) & 1u) ? c_make(1, 0) : c_make(-1, 0);
    return 0;
}

int uo_geometry(const ultra_hip_config* c, ultra_hip_geometry* g) {
    carriers cr;
    if (carriers_init(&cr, c) != 0) return ULTRA_HIP_ERR_INVALID_ARG;
    int k, m; code_params(c->code_rate, &k, &m);
    g->cp_len = cyclic_prefix(c);
    g->symbol_samples = c->fft_size + g->cp_len + c->symbol_guard;
    uint32_t tr = (c->entry == ULTRA_ENTRY_PRESYNCED) ? c->training_symbols : 0;
    g->frame_samples = (tr + c->n_data_symbols) * g->symbol_samples;
    g->n_data_carriers = (uint32_t)cr.n_data;
    g->n_pilot_carriers = (uint32_t)cr.n_pilot;
    g->bits_per_carrier = bits_per_symbol(c->modulation);
    g->llrs_per_symbol = g->n_data_carriers * g->bits_per_carrier;
    g->llrs_per_frame = g->llrs_per_symbol * c->n_data_symbols;
    g->ldpc_n = LDPC_N; g->ldpc_k = (uint32_t)k; g->ldpc_m = (uint32_t)m;
    g->ldpc_edges = (uint32_t)ldpc_get(c->code_rate)->edges;
    g->decoded_bytes = (uint32_t)((k + 7) / 8);
    return 0;
}

/* ====================================================================== */
/* Demodulator                                                             */
/* ====================================================================== */
typedef struct interp_info { int fft_idx, lower_pilot, upper_pilot; float alpha; } interp_info;

typedef struct demod {
    ultra_hip_config cfg;
    fft_plan fft;
    nco mixer;
    carriers cr;
    uint32_t cp, symbol_samples;
    int n_interp;
    interp_info interp[MAX_CARRIERS];

    /* per-frame state (src/ofdm/demodulator_impl.hpp:18-119) */
    cf channel_estimate[MAX_FFT];
    float noise_variance, estimated_snr_linear, snr_alpha;
    int snr_symbol_count;
    float freq_offset_hz, freq_offset_filtered, freq_correction_phase;
    int symbols_since_sync;
    int n_prev; cf prev_pilot_phases[MAX_CARRIERS];
    cf pilot_phase_correction;
    float timing_offset_samples;
    cf carrier_phase_correction; int carrier_phase_initialized;
    int n_lts; cf lts_carrier_phases[MAX_CARRIERS];
    int n_dprev; cf dbpsk_prev_equalized[MAX_CARRIERS];
    float carrier_noise_var[MAX_CARRIERS];
    /* adaptive equaliser (demodulator_impl.hpp:114-116; constructed :35-38 as (1,0) / (0,0) / 1.0) */
    cf lms_weights[MAX_FFT], last_decisions[MAX_FFT];
    float rls_P[MAX_FFT];

    /* output */
    float* soft; size_t n_soft, soft_cap;
} demod;

/* buildInterpTable, src/ofdm/demodulator.cpp:137-193 (ignores use_pilots) */
static void build_interp(demod* d) {
    const ultra_hip_config* c = &d->cfg;
    int fft_idx[MAX_CARRIERS + 2]; int is_pilot[MAX_CARRIERS + 2]; int nc = 0;
    int neg_limit = (int)(c->num_carriers / 2), pos_limit = (int)((c->num_carriers + 1) / 2);
    int pilot_count = 0;
    for (int i = -neg_limit; i <= pos_limit; ++i) {
        if (i == 0) continue;
        fft_idx[nc] = (int)(((unsigned)i + c->fft_size) % c->fft_size);
        is_pilot[nc] = (pilot_count % (int)c->pilot_spacing == 0);
        ++nc; ++pilot_count;
    }
    d->n_interp = 0;
    for (int ci = 0; ci < nc; ++ci) {
        if (is_pilot[ci]) continue;
        interp_info info; info.fft_idx = fft_idx[ci]; info.lower_pilot = -1; info.upper_pilot = -1; info.alpha = 0.5f;
        int lower_ci = -1, upper_ci = -1;
        for (int j = ci - 1; j >= 0; --j) if (is_pilot[j]) { info.lower_pilot = fft_idx[j]; lower_ci = j; break; }
        for (int j = ci + 1; j < nc; ++j) if (is_pilot[j]) { info.upper_pilot = fft_idx[j]; upper_ci = j; break; }
        if (lower_ci >= 0 && upper_ci >= 0) {
            float total_dist = (float)(upper_ci - lower_ci);
            info.alpha = (total_dist > 0) ? (float)(ci - lower_ci) / total_dist : 0.5f;
        }
        d->interp[d->n_interp++] = info;
    }
}

/* the adaptive equaliser's state as the constructor (demodulator.cpp:35-38), reset() (:1014-1016), the processPresynced
 * reset block (:894-897) and the mid-frame re-sync (:653-655) leave it */
static void adaptive_eq_reset(demod* d) {
    for (uint32_t i = 0; i < d->cfg.fft_size; ++i) {
        d->lms_weights[i] = c_make(1, 0); d->last_decisions[i] = c_make(0, 0); d->rls_P[i] = 1.0f;
    }
}
/* Impl::Impl, src/ofdm/demodulator.cpp:26-43 + member defaults demodulator_impl.hpp */
static int demod_init(demod* d, const ultra_hip_config* c) {
    memset(d, 0, sizeof(*d));
    d->cfg = *c;
    if (c->fft_size == 0 || (c->fft_size & (c->fft_size - 1)) != 0 || c->fft_size > MAX_FFT) return -1;
    if (c->pilot_spacing == 0) return -1;
    if (carriers_init(&d->cr, c) != 0) return -1;
    fft_init(&d->fft, c->fft_size);
    nco_init(&d->mixer, (float)c->center_freq, (float)c->sample_rate);
    d->cp = cyclic_prefix(c);
    d->symbol_samples = c->fft_size + d->cp + c->symbol_guard;
    for (uint32_t i = 0; i < c->fft_size; ++i) d->channel_estimate[i] = c_make(1, 0);
    d->noise_variance = 0.1f; d->estimated_snr_linear = 1.0f; d->snr_alpha = 0.3f;
    d->pilot_phase_correction = c_make(1, 0);
    d->carrier_phase_correction = c_make(1, 0);
    adaptive_eq_reset(d);
    build_interp(d);
    return 0;
}
static void demod_free(demod* d) { fft_free(&d->fft); }

/* Impl::toBaseband, src/ofdm/channel_equalizer.cpp:19-57 */
static void to_baseband(demod* d, const float* samples, size_t n, cf* bb) {
    float phase_increment = (float)(((double)-2.0f * M_PI * (double)d->freq_offset_hz) / (double)d->cfg.sample_rate);
    for (size_t i = 0; i < n; ++i) {
        cf osc = nco_next(&d->mixer);
        cf cj = c_conj(osc);
        cf mixed = c_make(cj.re * samples[i], cj.im * samples[i]); /* float * complex */
        if (fabsf(d->freq_offset_hz) > 0.01f) {
            cf corr = c_make(cosf(d->freq_correction_phase), sinf(d->freq_correction_phase));
            mixed = c_mul(mixed, corr);
            d->freq_correction_phase += phase_increment;
            if ((double)d->freq_correction_phase > M_PI)
                d->freq_correction_phase = (float)((double)d->freq_correction_phase - (double)2.0f * M_PI);
            else if ((double)d->freq_correction_phase < -M_PI)
                d->freq_correction_phase = (float)((double)d->freq_correction_phase + (double)2.0f * M_PI);
        }
        bb[i] = mixed;
    }
}

/* Impl::extractSymbol, src/ofdm/channel_equalizer.cpp:59-71 */
static void extract_symbol(demod* d, const cf* bb, size_t nbb, cf* freq) {
    size_t start = d->cp;
    for (size_t i = 0; i < d->cfg.fft_size; ++i) freq[i] = (start + i < nbb) ? bb[start + i] : c_make(0, 0);
    fft_exec(&d->fft, freq, 0);
}

static int is_differential(uint32_t mod) {
    return mod == ULTRA_MOD_DBPSK || mod == ULTRA_MOD_DQPSK || mod == ULTRA_MOD_D8PSK;
}

/* Impl::interpolateChannel, src/ofdm/channel_equalizer.cpp:601-631 */
static void interpolate_channel(demod* d) {
    for (int dc = 0; dc < d->n_interp; ++dc) {
        const interp_info* info = &d->interp[dc];
        if (info->lower_pilot >= 0 && info->upper_pilot >= 0) {
            cf H1 = d->channel_estimate[info->lower_pilot];
            cf H2 = d->channel_estimate[info->upper_pilot];
            cf pd = c_mul(H2, c_conj(H1));
            float phase_diff = fabsf(atan2f(pd.im, pd.re));
            if (phase_diff > 1.5708f) {
                d->channel_estimate[info->fft_idx] = (info->alpha < 0.5f) ? H1 : H2;
            } else {
                d->channel_estimate[info->fft_idx] =
                    c_add(c_scale(H1, 1.0f - info->alpha), c_scale(H2, info->alpha));
            }
        } else if (info->lower_pilot >= 0) {
            d->channel_estimate[info->fft_idx] = d->channel_estimate[info->lower_pilot];
        } else if (info->upper_pilot >= 0) {
            d->channel_estimate[info->fft_idx] = d->channel_estimate[info->upper_pilot];
        }
    }
}

static inline int wrap_k(int idx, uint32_t fft) {
    int k = idx;
    if (k > (int)fft / 2) k -= (int)fft;
    return k;
}
/* float timing_phase = 2.0f * M_PI * k * timing_offset_samples / config.fft_size; */
static inline float timing_phase_of(int k, float timing, uint32_t fft) {
    return (float)((((double)2.0f * M_PI * (double)k) * (double)timing) / (double)fft);
}

/* Impl::updateChannelEstimate, src/ofdm/channel_equalizer.cpp:330-595 */
static void update_channel_estimate(demod* d, const cf* freq) {
    const carriers* cr = &d->cr;
    const int np = cr->n_pilot;
    float alpha = (d->snr_symbol_count == 0) ? 1.0f : 0.9f;

    cf h_ls_all[MAX_CARRIERS];
    cf h_sum = c_make(0, 0);
    for (int i = 0; i < np; ++i) {
        h_ls_all[i] = c_div(freq[cr->pilot_idx[i]], cr->pilot_seq[i]);
        h_sum = c_add(h_sum, h_ls_all[i]);
    }
    /* carrier phase recovery on the first symbol (:348-357) */
    if (!d->carrier_phase_initialized && np != 0) {
        cf h_avg = c_divf(h_sum, (float)np);
        float avg_mag = c_abs(h_avg);
        if (avg_mag > 0.01f) {
            d->carrier_phase_correction = c_divf(c_conj(h_avg), avg_mag);
            d->carrier_phase_initialized = 1;
        }
    }
    for (int i = 0; i < np; ++i) h_ls_all[i] = c_mul(h_ls_all[i], d->carrier_phase_correction);
    h_sum = c_mul(h_sum, d->carrier_phase_correction);

    /* signal power (:385-389); 0/0 = NaN when there are no pilots (quirk 4) */
    float signal_power_sum = 0.0f;
    for (int i = 0; i < np; ++i) signal_power_sum += c_norm(h_ls_all[i]);
    float signal_power = signal_power_sum / (float)(size_t)np;

    /* temporal noise + smoothed H (:391-412) */
    float noise_power_sum = 0.0f;
    size_t noise_count = 0;
    for (int i = 0; i < np; ++i) {
        int idx = cr->pilot_idx[i];
        if (d->n_prev != 0 && i < d->n_prev) {
            cf prev_h = d->prev_pilot_phases[i], curr_h = h_ls_all[i];
            if (c_norm(prev_h) > 1e-6f && c_norm(curr_h) > 1e-6f) {
                cf diff = c_sub(curr_h, prev_h);
                noise_power_sum += c_norm(diff);
                noise_count++;
            }
        }
        cf h_old = d->channel_estimate[idx];
        d->channel_estimate[idx] = c_add(c_scale(h_ls_all[i], alpha), c_scale(h_old, 1.0f - alpha));
    }
    if (noise_count == 0) { noise_power_sum = signal_power / 31.6f; noise_count = 1; }

    /* CFO from pilot phase differences (:420-470) */
    if (d->n_prev != 0 && d->n_prev == np) {
        cf phase_diff_sum = c_make(0, 0);
        int valid_count = 0;
        for (int i = 0; i < np; ++i) {
            cf diff = c_mul(h_ls_all[i], c_conj(d->prev_pilot_phases[i]));
            if (c_norm(d->prev_pilot_phases[i]) > 1e-6f && c_norm(h_ls_all[i]) > 1e-6f) {
                float mag = c_abs(diff);
                if (mag > 1e-6f) { phase_diff_sum = c_add(phase_diff_sum, c_divf(diff, mag)); valid_count++; }
            }
        }
        if (valid_count > 0) {
            cf avg_diff = c_divf(phase_diff_sum, (float)valid_count);
            float avg_phase_diff = atan2f(avg_diff.im, avg_diff.re);
            d->pilot_phase_correction = c_make(cosf(-avg_phase_diff), sinf(-avg_phase_diff));
            float symbol_duration = (float)d->symbol_samples / (float)d->cfg.sample_rate;
            float residual_cfo = (float)((double)avg_phase_diff / ((double)2.0f * M_PI * (double)symbol_duration));
            float total_cfo = d->freq_offset_hz + residual_cfo;
            float adaptive_alpha = 0.3f;
            if (d->symbols_since_sync < 10) {
                float progress = (float)d->symbols_since_sync / 10;
                adaptive_alpha = 0.9f * (1.0f - progress) + 0.3f * progress;
            }
            if (fabsf(residual_cfo) > 10.0f) adaptive_alpha = f_max(adaptive_alpha, 0.9f);
            d->symbols_since_sync++;
            d->freq_offset_filtered = adaptive_alpha * total_cfo + (1.0f - adaptive_alpha) * d->freq_offset_filtered;
            d->freq_offset_hz = f_max(-90.0f, f_min(90.0f, d->freq_offset_filtered));
        }
    } else {
        d->pilot_phase_correction = c_make(1, 0);
    }

    /* timing from pilot phase slope (:472-509) */
    if (d->snr_symbol_count >= 3) {
        float sum_k = 0, sum_k2 = 0, sum_phase = 0, sum_k_phase = 0;
        int tv = 0;
        for (int i = 0; i < np; ++i) {
            if (c_norm(h_ls_all[i]) < 1e-6f) continue;
            int k = wrap_k(cr->pilot_idx[i], d->cfg.fft_size);
            float phase = c_arg(h_ls_all[i]);
            sum_k += (float)k;
            sum_k2 += (float)(k * k);
            sum_phase += phase;
            sum_k_phase += (float)k * phase;
            tv++;
        }
        if (tv >= 3) {
            float n = (float)tv;
            float denom = n * sum_k2 - sum_k * sum_k;
            if (fabsf(denom) > 1e-6f) {
                float slope = (n * sum_k_phase - sum_k * sum_phase) / denom;
                float inst = (float)((double)(slope * (float)d->cfg.fft_size) / ((double)2.0f * M_PI));
                d->timing_offset_samples = 0.3f * inst + (1.0f - 0.3f) * d->timing_offset_samples;
                float max_timing = 50.0f * ((float)d->cfg.fft_size / 512.0f);
                d->timing_offset_samples = f_max(-max_timing, f_min(max_timing, d->timing_offset_samples));
            }
        }
    }

    for (int i = 0; i < np; ++i) d->prev_pilot_phases[i] = h_ls_all[i];
    d->n_prev = np;

    /* coherent timing fix around interpolation (:514-567) */
    int coherent = !is_differential(d->cfg.modulation);
    if (coherent && fabsf(d->timing_offset_samples) > 0.1f) {
        for (int i = 0; i < np; ++i) {
            int idx = cr->pilot_idx[i];
            float tp = timing_phase_of(wrap_k(idx, d->cfg.fft_size), d->timing_offset_samples, d->cfg.fft_size);
            d->channel_estimate[idx] = c_mul(d->channel_estimate[idx], c_expj(-tp));
        }
    }
    interpolate_channel(d);
    if (coherent && fabsf(d->timing_offset_samples) > 0.1f) {
        for (int i = 0; i < np; ++i) {
            int idx = cr->pilot_idx[i];
            float tp = timing_phase_of(wrap_k(idx, d->cfg.fft_size), d->timing_offset_samples, d->cfg.fft_size);
            d->channel_estimate[idx] = c_mul(d->channel_estimate[idx], c_expj(tp));
        }
        for (int i = 0; i < cr->n_data; ++i) {
            int idx = cr->data_idx[i];
            float tp = timing_phase_of(wrap_k(idx, d->cfg.fft_size), d->timing_offset_samples, d->cfg.fft_size);
            d->channel_estimate[idx] = c_mul(d->channel_estimate[idx], c_expj(tp));
        }
    }

    /* adaptive equaliser weights start from the pilot-based estimate during the first three symbols (:569-581) */
    if (d->cfg.adaptive_eq_enabled && d->snr_symbol_count < 3) {
        for (int i = 0; i < cr->n_data; ++i) d->lms_weights[cr->data_idx[i]] = d->channel_estimate[cr->data_idx[i]];
        for (int i = 0; i < cr->n_pilot; ++i) d->lms_weights[cr->pilot_idx[i]] = d->channel_estimate[cr->pilot_idx[i]];
    }

    /* noise variance + SNR EMA (:583-592) */
    if (noise_count > 1 && noise_power_sum > 0.0f) {
        d->noise_variance = noise_power_sum / (float)(noise_count - 1);
        if (d->noise_variance < 1e-6f) d->noise_variance = 1e-6f;
        float inst_snr = signal_power / d->noise_variance;
        inst_snr = f_max(0.1f, f_min(10000.0f, inst_snr));
        d->estimated_snr_linear = d->snr_alpha * inst_snr + (1.0f - d->snr_alpha) * d->estimated_snr_linear;
    }
    d->snr_symbol_count++;
}

/* Impl::hardDecision, src/ofdm/channel_equalizer.cpp:637-700 */
static float slice_qam16(float x) { return (x < -0.4f) ? -0.9487f : (x < 0.0f) ? -0.3162f : (x < 0.4f) ? 0.3162f : 0.9487f; }
static float slice8(float x, float d) {
    if (x < -6 * d) return -7 * d;
    if (x < -4 * d) return -5 * d;
    if (x < -2 * d) return -3 * d;
    if (x < 0) return -d;
    if (x < 2 * d) return d;
    if (x < 4 * d) return 3 * d;
    if (x < 6 * d) return 5 * d;
    return 7 * d;
}
static cf hard_decision(cf sym, uint32_t mod) {
    switch (mod) {
        case ULTRA_MOD_BPSK: return c_make(sym.re > 0 ? 1.0f : -1.0f, 0);
        case ULTRA_MOD_QAM16: return c_make(slice_qam16(sym.re), slice_qam16(sym.im));
        case ULTRA_MOD_QAM32: {
            const float d = 0.1961161351381840f;   /* QAM32_SCALE, demodulator_constants.hpp:90 */
            float I = (sym.re < -2 * d) ? -3 * d : (sym.re < 0) ? -d : (sym.re < 2 * d) ? d : 3 * d;
            return c_make(I, slice8(sym.im, d));
        }
        case ULTRA_MOD_QAM64: return c_make(slice8(sym.re, 0.1543f), slice8(sym.im, 0.1543f));
        case ULTRA_MOD_QPSK:
        default: return c_make(sym.re > 0 ? 0.7071f : -0.7071f, sym.im > 0 ? 0.7071f : -0.7071f);
    }
}
/* Impl::lmsUpdate / rlsUpdate, src/ofdm/channel_equalizer.cpp:705-722 */
static void lms_update(demod* d, int idx, cf received, cf reference) {
    float mu = d->cfg.lms_mu;
    cf error = c_sub(received, c_mul(d->lms_weights[idx], reference));
    d->lms_weights[idx] = c_add(d->lms_weights[idx], c_mul(c_scale(c_conj(reference), mu), error));
}
static void rls_update(demod* d, int idx, cf received, cf reference) {
    float lambda = d->cfg.rls_lambda;
    float P = d->rls_P[idx];
    float ref_norm = c_norm(reference);
    float k = P / (lambda + P * ref_norm);
    cf error = c_sub(received, c_mul(d->lms_weights[idx], reference));
    d->lms_weights[idx] = c_add(d->lms_weights[idx], c_mul(c_scale(c_conj(reference), k), error));
    d->rls_P[idx] = (P - k * ref_norm * P) / lambda;
    d->rls_P[idx] = f_max(0.001f, f_min(1000.0f, d->rls_P[idx]));   /* ADAPTIVE_EQ_P_MIN / _MAX */
}

/* Impl::equalize, src/ofdm/channel_equalizer.cpp:728-840 */
static void equalize(demod* d, const cf* freq, cf* eq) {
    const carriers* cr = &d->cr;
    const int nd = cr->n_data;
    uint32_t mod = d->cfg.modulation;
    if (is_differential(mod)) {
        for (int i = 0; i < nd; ++i) {
            int idx = cr->data_idx[i];
            cf received = freq[idx], h = d->channel_estimate[idx];
            float h_power = c_norm(h);
            float tp = timing_phase_of(wrap_k(idx, d->cfg.fft_size), d->timing_offset_samples, d->cfg.fft_size);
            cf tc = c_expj(tp);
            if (h_power > 1e-6f) {
                cf t = c_divf(c_mul(received, c_conj(h)), h_power);
                eq[i] = c_mul(c_mul(t, d->pilot_phase_correction), tc);
                d->carrier_noise_var[i] = d->noise_variance / h_power;
            } else {
                eq[i] = c_mul(c_mul(received, d->pilot_phase_correction), tc);
                d->carrier_noise_var[i] = 100.0f;
            }
            d->carrier_noise_var[i] = f_max(1e-6f, f_min(100.0f, d->carrier_noise_var[i]));
        }
        return;
    }
    for (int i = 0; i < nd; ++i) {
        int idx = cr->data_idx[i];
        if (d->cfg.adaptive_eq_enabled) {                  /* :779-805 — no clamp of the noise variance on this branch */
            cf received = freq[idx], h = d->lms_weights[idx];
            float h_power = c_norm(h);
            float mmse_denom = h_power + d->noise_variance;
            if (mmse_denom < 1e-10f) {
                eq[i] = c_make(0, 0);
                d->carrier_noise_var[i] = 100.0f;
            } else {
                eq[i] = c_divf(c_mul(c_conj(h), received), mmse_denom);
                d->carrier_noise_var[i] = d->noise_variance / (h_power + 1e-6f);
            }
            if (d->cfg.decision_directed) {
                cf decision = hard_decision(eq[i], mod);
                if (d->cfg.adaptive_eq_use_rls) rls_update(d, idx, received, decision);
                else lms_update(d, idx, received, decision);
                d->last_decisions[idx] = decision;
            }
            continue;
        }
        cf received = freq[idx], h = d->channel_estimate[idx];
        float h_power = c_norm(h);
        float mmse_denom = h_power + d->noise_variance;
        if (mmse_denom < 1e-10f) {
            eq[i] = c_make(0, 0);
            d->carrier_noise_var[i] = 100.0f;
        } else {
            eq[i] = c_divf(c_mul(c_conj(h), received), mmse_denom);
            d->carrier_noise_var[i] = d->noise_variance / (h_power + 1e-6f);
            d->carrier_noise_var[i] = f_max(1e-6f, f_min(100.0f, d->carrier_noise_var[i]));
        }
    }
    /* deep-fade soft erasure (:822-837) */
    float avg_h_power = 0.0f;
    for (int i = 0; i < nd; ++i) avg_h_power += c_norm(d->channel_estimate[cr->data_idx[i]]);
    avg_h_power /= (float)(size_t)nd;
    float fade_threshold = 0.1f * avg_h_power;
    for (int i = 0; i < nd; ++i) {
        float h_power = c_norm(d->channel_estimate[cr->data_idx[i]]);
        if (h_power < fade_threshold) d->carrier_noise_var[i] = 100.0f;
    }
}

/* soft_demap::clipLLR, src/ofdm/soft_demap.hpp:22-29 */
static inline float clip_llr(float llr) {
    float clipped = f_max(-10.0f, f_min(10.0f, llr));
    if (fabsf(clipped) < 0.5f) clipped = (clipped >= 0) ? 0.5f : -0.5f;
    return clipped;
}
static inline void push_soft(demod* d, float v) { if (d->n_soft < d->soft_cap) d->soft[d->n_soft] = v; d->n_soft++; }

/* CE margins, src/ofdm/soft_demap.hpp:243-264 + demodulator_constants.hpp:102-107 */
static float ce_margin(uint32_t mod) {
    switch (mod) {
        case ULTRA_MOD_DBPSK: case ULTRA_MOD_DQPSK: case ULTRA_MOD_BPSK: case ULTRA_MOD_QPSK: return 1.0f;
        case ULTRA_MOD_D8PSK: case ULTRA_MOD_QAM8: return 1.1f;
        case ULTRA_MOD_QAM16: return 1.2f;
        case ULTRA_MOD_QAM32: return 1.5f;
        case ULTRA_MOD_QAM64: return 1.8f;
        case ULTRA_MOD_QAM256: return 2.5f;
        default: return 1.0f;
    }
}

static void demap_qpsk(demod* d, cf sym, float nv) { /* soft_demap.hpp:42-45 */
    float scale = (-2.0f * 0.7071067811865476f) / nv;
    push_soft(d, clip_llr(sym.re * scale));
    push_soft(d, clip_llr(sym.im * scale));
}

/* Impl::demodulateSymbol, src/ofdm/demodulator.cpp:199-435 (GUI ring omitted) */
static void demodulate_symbol(demod* d, const cf* eq) {
    const int nd = d->cr.n_data;
    uint32_t mod = d->cfg.modulation;
    float margin = ce_margin(mod);

    if ((mod == ULTRA_MOD_DQPSK || mod == ULTRA_MOD_D8PSK) && d->n_dprev == 0) {
        d->n_dprev = nd;
        if (d->n_lts != 0 && d->n_lts == nd) {
            for (int i = 0; i < nd; ++i) d->dbpsk_prev_equalized[i] = d->lts_carrier_phases[i];
        } else { /* sync_sequence.size() == num_carriers >= nd always: (1,0) fallback (:260-265) */
            for (int i = 0; i < nd; ++i) d->dbpsk_prev_equalized[i] = c_make(1, 0);
        }
    }

    for (int i = 0; i < nd; ++i) {
        cf sym = eq[i];
        float nv = d->carrier_noise_var[i] * margin;
        switch (mod) {
            case ULTRA_MOD_DBPSK: { /* soft_demap.hpp:173-187 */
                if (d->n_dprev == 0) { d->n_dprev = nd; for (int q = 0; q < nd; ++q) d->dbpsk_prev_equalized[q] = c_make(1, 0); }
                cf prev = d->dbpsk_prev_equalized[i];
                cf diff = c_mul(sym, c_conj(prev));
                float phase_diff = atan2f(diff.im, diff.re);
                float sp = c_abs(sym) * c_abs(prev);
                float llr;
                if (sp < 1e-6f) llr = 0.0f;
                else llr = clip_llr(2.0f * sp * cosf(phase_diff) / nv);
                push_soft(d, llr);
                d->dbpsk_prev_equalized[i] = sym;
                break;
            }
            case ULTRA_MOD_DQPSK: { /* soft_demap.hpp:192-213 */
                cf prev = d->dbpsk_prev_equalized[i];
                cf diff = c_mul(sym, c_conj(prev));
                float phase = atan2f(diff.im, diff.re);
                float sp = c_abs(sym) * c_abs(prev);
                if (sp < 1e-6f) { push_soft(d, 0.0f); push_soft(d, 0.0f); }
                else {
                    float scale = 2.0f * sp / nv;
                    const float pi = 3.14159265358979f;
                    push_soft(d, clip_llr(scale * sinf(phase + pi / 4)));
                    push_soft(d, clip_llr(scale * cosf(2 * phase)));
                }
                d->dbpsk_prev_equalized[i] = sym;
                break;
            }
            case ULTRA_MOD_D8PSK: { /* soft_demap.hpp:217-237 */
                cf prev = d->dbpsk_prev_equalized[i];
                cf diff = c_mul(sym, c_conj(prev));
                float pd = atan2f(diff.im, diff.re);
                float sp = c_abs(sym) * c_abs(prev);
                if (sp < 1e-6f) { push_soft(d, 0.0f); push_soft(d, 0.0f); push_soft(d, 0.0f); }
                else {
                    float conf = sp / nv;
                    push_soft(d, clip_llr(conf * sinf(pd)));
                    push_soft(d, clip_llr(conf * sinf(2.0f * pd)));
                    push_soft(d, clip_llr(conf * sinf(4.0f * pd)));
                }
                d->dbpsk_prev_equalized[i] = sym;
                break;
            }
            case ULTRA_MOD_BPSK: /* soft_demap.hpp:37-39 */
                push_soft(d, clip_llr(-2.0f * sym.re / nv));
                break;
            case ULTRA_MOD_QPSK:
                demap_qpsk(d, sym, nv);
                break;
            case ULTRA_MOD_QAM16: { /* soft_demap.hpp:49-64 */
                float I = sym.re, Q = sym.im, scale = 2.0f / nv;
                push_soft(d, clip_llr(-scale * I));
                push_soft(d, clip_llr(scale * (fabsf(I) - 0.6324555320336759f)));
                push_soft(d, clip_llr(-scale * Q));
                push_soft(d, clip_llr(scale * (fabsf(Q) - 0.6324555320336759f)));
                break;
            }
            case ULTRA_MOD_QAM32: { /* soft_demap.hpp:68-121, max-log over 32 points */
                static const float I_LEVELS[4] = {-3, -1, 1, 3};
                static const int I_GRAY[4] = {0, 1, 3, 2};
                static const float Q_LEVELS[8] = {-7, -5, -3, -1, 1, 3, 5, 7};
                static const int Q_GRAY[8] = {0, 1, 3, 2, 6, 7, 5, 4};
                const float S = 0.1961161351381840f;
                float sf = 2.0f / nv;
                for (int b = 0; b < 5; ++b) {
                    int mask = 1 << (4 - b);
                    float m0 = 1e10f, m1 = 1e10f;
                    for (int qi = 0; qi < 8; ++qi) for (int ii = 0; ii < 4; ++ii) {
                        cf pos = c_make(I_LEVELS[ii] * S, Q_LEVELS[qi] * S);
                        int bits = (Q_GRAY[qi] << 2) | I_GRAY[ii];
                        cf df = c_sub(sym, pos);
                        float dist = df.re * df.re + df.im * df.im;
                        if (bits & mask) { if (dist < m1) m1 = dist; }
                        else { if (dist < m0) m0 = dist; }
                    }
                    push_soft(d, clip_llr(sf * (m1 - m0)));
                }
                break;
            }
            case ULTRA_MOD_QAM64: { /* soft_demap.hpp:124-141 */
                float I = sym.re, Q = sym.im, scale = 2.0f / nv;
                const float D2 = 0.3086067f, D4 = 0.6172134f;
                push_soft(d, clip_llr(-scale * I));
                push_soft(d, clip_llr(scale * (fabsf(I) - D4)));
                push_soft(d, clip_llr(scale * (fabsf(fabsf(I) - D4) - D2)));
                push_soft(d, clip_llr(-scale * Q));
                push_soft(d, clip_llr(scale * (fabsf(Q) - D4)));
                push_soft(d, clip_llr(scale * (fabsf(fabsf(Q) - D4) - D2)));
                break;
            }
            case ULTRA_MOD_QAM256: { /* soft_demap.hpp:144-163 */
                float I = sym.re, Q = sym.im, scale = 2.0f / nv;
                const float D2 = 0.1290994f, D4 = 0.2581989f, D8 = 0.5163978f;
                push_soft(d, clip_llr(-scale * I));
                push_soft(d, clip_llr(scale * (fabsf(I) - D8)));
                push_soft(d, clip_llr(scale * (fabsf(fabsf(I) - D8) - D4)));
                push_soft(d, clip_llr(scale * (fabsf(fabsf(fabsf(I) - D8) - D4) - D2)));
                push_soft(d, clip_llr(-scale * Q));
                push_soft(d, clip_llr(scale * (fabsf(Q) - D8)));
                push_soft(d, clip_llr(scale * (fabsf(fabsf(Q) - D8) - D4)));
                push_soft(d, clip_llr(scale * (fabsf(fabsf(fabsf(Q) - D8) - D4) - D2)));
                break;
            }
            default:
                demap_qpsk(d, sym, nv);
        }
    }

    /* decision-directed tracking, differential modes (:362-434).  Note the
     * loop above already stored dbpsk_prev_equalized[i] = equalized[i], so
     * prev_sym here IS the current symbol — restated literally. */
    if ((mod == ULTRA_MOD_DQPSK || mod == ULTRA_MOD_D8PSK) && d->n_dprev != 0) {
        if (d->snr_symbol_count >= 1) {
            cf phase_error_sum = c_make(0, 0);
            int valid_count = 0;
            float dd_alpha = (d->snr_symbol_count < 3) ? 0.3f : 0.15f;
            for (int i = 0; i < nd; ++i) {
                int idx = d->cr.data_idx[i];
                cf prev = (i < d->n_dprev) ? d->dbpsk_prev_equalized[i] : c_make(1, 0);
                float sp = c_abs(eq[i]) * c_abs(prev);
                if (sp > 0.1f) {
                    cf diff = c_mul(eq[i], c_conj(prev));
                    float phase = atan2f(diff.im, diff.re);
                    float expected;
                    if (mod == ULTRA_MOD_DQPSK) {
                        int quadrant = (int)round((double)(phase * 2.0f) / M_PI);
                        quadrant = ((quadrant % 4) + 4) % 4;
                        expected = (float)((double)quadrant * M_PI / (double)2.0f);
                    } else {
                        int octant = (int)round((double)(phase * 4.0f) / M_PI);
                        octant = ((octant % 8) + 8) % 8;
                        expected = (float)((double)octant * M_PI / (double)4.0f);
                    }
                    float phase_error = phase - expected;
                    while ((double)phase_error > M_PI) phase_error = (float)((double)phase_error - 2 * M_PI);
                    while ((double)phase_error < -M_PI) phase_error = (float)((double)phase_error + 2 * M_PI);
                    float max_err = (mod == ULTRA_MOD_DQPSK) ? 0.7f : 0.35f;
                    if (fabsf(phase_error) < max_err) {
                        cf pc = c_make(cosf(-phase_error * dd_alpha), sinf(-phase_error * dd_alpha));
                        d->channel_estimate[idx] = c_mul(d->channel_estimate[idx], pc);
                    }
                    cf e = c_make(cosf(phase_error), sinf(phase_error));
                    phase_error_sum = c_add(phase_error_sum, c_make(e.re * sp, e.im * sp));
                    valid_count++;
                }
            }
            if (valid_count >= 5) {
                float avg = atan2f(phase_error_sum.im, phase_error_sum.re);
                cf corr = c_make(cosf(-avg), sinf(-avg));
                float a = (d->snr_symbol_count < 5) ? 0.5f : 0.2f;
                float ang = a * c_arg(corr);
                cf t = c_scale(d->pilot_phase_correction, powf(c_abs(corr), a));
                d->pilot_phase_correction = c_mul(t, c_make(cosf(ang), sinf(ang)));
                float mag = c_abs(d->pilot_phase_correction);
                if (mag > 0.01f) d->pilot_phase_correction = c_divf(d->pilot_phase_correction, mag);
            }
        }
    }
}

/* Impl::estimateChannelFromLTS, src/ofdm/channel_equalizer.cpp:77-328 */
static void estimate_channel_from_lts(demod* d, const float* training, size_t num_symbols) {
    const carriers* cr = &d->cr;
    const int nd = cr->n_data, np = cr->n_pilot;
    if (num_symbols == 0 || nd == 0) return;
    cf h_last[MAX_CARRIERS], h_sum_pilot[MAX_CARRIERS];
    for (int i = 0; i < nd; ++i) h_last[i] = c_make(0, 0);
    for (int i = 0; i < np; ++i) h_sum_pilot[i] = c_make(0, 0);
    size_t valid = 0;
    cf* bb = (cf*)malloc(sizeof(cf) * d->symbol_samples);
    cf freq[MAX_FFT];
    const float* ptr = training;
    for (size_t sym = 0; sym < num_symbols; ++sym) {
        to_baseband(d, ptr, d->symbol_samples, bb);
        extract_symbol(d, bb, d->symbol_samples, freq);
        for (int i = 0; i < nd; ++i) {
            cf rx = freq[cr->data_idx[i]], tx = cr->sync_seq[i % cr->n_sync];
            cf h = c_make(0, 0);                 /* h_per_symbol default-constructed */
            if (c_abs(tx) > 0.01f) h = c_div(rx, tx);
            if (sym == num_symbols - 1) h_last[i] = h;
        }
        for (int i = 0; i < np; ++i) {
            cf rx = freq[cr->pilot_idx[i]], tx = cr->pilot_seq[i];
            if (c_abs(tx) > 0.01f) h_sum_pilot[i] = c_add(h_sum_pilot[i], c_div(rx, tx));
        }
        valid++;
        ptr += d->symbol_samples;
    }
    free(bb);
    for (int i = 0; i < nd; ++i) d->channel_estimate[cr->data_idx[i]] = h_last[i];
    float inv_count = 1.0f / (float)valid;
    for (int i = 0; i < np; ++i) d->channel_estimate[cr->pilot_idx[i]] = c_scale(h_sum_pilot[i], inv_count);

    float h_mag_sum = 0;
    for (int i = 0; i < nd; ++i) h_mag_sum += c_abs(d->channel_estimate[cr->data_idx[i]]);
    float h_mag_avg = h_mag_sum / (float)(size_t)nd;
    if (h_mag_avg > 1e-6f && d->noise_variance > 1e-10f) {
        float sp = h_mag_avg * h_mag_avg;
        d->estimated_snr_linear = sp / d->noise_variance;
        d->estimated_snr_linear = f_max(0.1f, f_min(10000.0f, d->estimated_snr_linear));
    }
    d->n_lts = nd;
    for (int i = 0; i < nd; ++i) d->lts_carrier_phases[i] = c_make(1.0f, 0.0f);
    d->snr_symbol_count = (int)num_symbols;
}

/* SYNCED-entry state (src/ofdm/demodulator.cpp:533-591 on a fresh object) */
static void demod_enter_synced(demod* d, float cfo_hz) {
    d->freq_offset_hz = cfo_hz;
    d->freq_offset_filtered = cfo_hz;
    d->freq_correction_phase = 0.0f;
    d->symbols_since_sync = 0;
    d->mixer.phase = 0;
    d->n_dprev = 0;
    d->carrier_phase_initialized = 0;
    d->carrier_phase_correction = c_make(1, 0);
    d->timing_offset_samples = 0.0f;
}

/* one pass of the SYNCED symbol loop body, src/ofdm/demodulator.cpp:672-697 */
static void demod_symbol(demod* d, const float* samples, cf* bb, cf* freq, cf* eq, int with_update) {
    to_baseband(d, samples, d->symbol_samples, bb);
    extract_symbol(d, bb, d->symbol_samples, freq);
    if (with_update) update_channel_estimate(d, freq);
    equalize(d, freq, eq);
    demodulate_symbol(d, eq);
}

static void dump_scal(const demod* d, float* s) {
    s[0] = d->freq_offset_hz; s[1] = d->noise_variance; s[2] = d->estimated_snr_linear;
    s[3] = d->timing_offset_samples; s[4] = d->freq_correction_phase;
    s[5] = d->pilot_phase_correction.re; s[6] = d->pilot_phase_correction.im;
    s[7] = (float)d->snr_symbol_count;
}

int uo_demod_synced(const ultra_hip_config* c, const float* audio, uint32_t n_symbols, float cfo_hz,
                    float* llr_out, uint32_t llr_cap, float* stage_out) {
    demod* d = (demod*)malloc(sizeof(demod));
    if (demod_init(d, c) != 0) { free(d); return -1; }
    demod_enter_synced(d, cfo_hz);
    d->soft = llr_out; d->soft_cap = llr_cap; d->n_soft = 0;
    const size_t S = d->symbol_samples, N = c->fft_size;
    const int nd = d->cr.n_data;
    cf* bb = (cf*)malloc(sizeof(cf) * S);
    cf freq[MAX_FFT], eq[MAX_CARRIERS];
    float* st = stage_out;
    for (uint32_t s = 0; s < n_symbols; ++s) {
        demod_symbol(d, audio + (size_t)s * S, bb, freq, eq, 1);
        if (st) {
            memcpy(st, bb, sizeof(cf) * S); st += 2 * S;
            memcpy(st, freq, sizeof(cf) * N); st += 2 * N;
            memcpy(st, d->channel_estimate, sizeof(cf) * N); st += 2 * N;
            memcpy(st, eq, sizeof(cf) * (size_t)nd); st += 2 * nd;
            memcpy(st, d->carrier_noise_var, sizeof(float) * (size_t)nd); st += nd;
            dump_scal(d, st); st += 8;
        }
    }
    int n = (int)d->n_soft;
    free(bb); demod_free(d); free(d);
    return n;
}

/* OFDMDemodulator::Impl::estimateCFOFromTraining(samples, num_symbols, coarse_cfo_hz = 0), src/ofdm/ofdm_sync.cpp:278-380:
 * the first two training symbols mixed to baseband by a FRESH NCO(center_freq), correlated FFT part against FFT part;
 * CFO = arg(P) fs / (2 pi sym_len), 0 when the normalised correlation is below 0.3, clamped to +-fs / (2 sym_len). */
static float estimate_cfo_from_training(const demod* d, const float* samples, size_t num_symbols) {
    if (num_symbols < 2) return 0.0f;
    const ultra_hip_config* c = &d->cfg;
    const size_t fft_len = c->fft_size, cp_len = d->cp, sym_len = d->symbol_samples, total = 2 * sym_len;
    nco mix;
    nco_init(&mix, (float)c->center_freq, (float)c->sample_rate);
    cf* bb = (cf*)malloc(sizeof(cf) * total);
    for (size_t i = 0; i < total; ++i) {
        cf osc = nco_next(&mix);
        bb[i] = c_make(samples[i] * osc.re, samples[i] * -osc.im);        /* float * conj(osc); coarse_cfo = 0: no rotation */
    }
    cf P = c_make(0.0f, 0.0f);
    float E1 = 0.0f, E2 = 0.0f;
    const size_t s1 = cp_len, s2 = sym_len + cp_len;
    for (size_t i = 0; i < fft_len; ++i) {
        if (s1 + i < total && s2 + i < total) {
            cf z1 = bb[s1 + i], z2 = bb[s2 + i];
            P = c_add(P, c_mul(c_conj(z1), z2));
            E1 += c_norm(z1);
            E2 += c_norm(z2);
        }
    }
    free(bb);
    float corr_mag = c_abs(P) / sqrtf(E1 * E2 + 1e-10f);
    if (corr_mag < 0.3f) return 0.0f;
    float phase = atan2f(P.im, P.re);
    /* phase * config.sample_rate / (2.0f * M_PI * sym_len): float * (float)uint32, then a double division */
    float cfo_hz = (float)((double)(phase * (float)c->sample_rate) / ((double)2.0f * M_PI * (double)sym_len));
    float max_cfo = (float)c->sample_rate / (2.0f * (float)sym_len);
    return f_max(-max_cfo, f_min(max_cfo, cfo_hz));
}

/* OFDMDemodulator::processPresynced, src/ofdm/demodulator.cpp:854-985.  cfo_hz finite: after
 * setFrequencyOffset[WithPhase] (:805-825, the chirp-CFO-trusted branch :918-919).  cfo_hz NaN: the frequency offset
 * was never set on this demodulator — with two or more training symbols the CFO comes from estimateCFOFromTraining
 * (:920-925), else it stays 0 (:926-928); the correction phase starts at 0. */
static int demod_presynced_run(demod* d, const float* audio, size_t n_samples, float cfo_hz, float cfo_phase,
                               cf* bb, cf* freq, cf* eq) {
    const ultra_hip_config* c = &d->cfg;
    if (n_samples < d->symbol_samples) return 0;
    if (cfo_hz != cfo_hz) {
        cfo_phase = 0.0f;
        cfo_hz = (c->training_symbols >= 2 && n_samples >= 2 * (size_t)d->symbol_samples)
                     ? estimate_cfo_from_training(d, audio, c->training_symbols) : 0.0f;
    }
    d->freq_offset_hz = cfo_hz; d->freq_offset_filtered = cfo_hz; d->freq_correction_phase = cfo_phase;
    d->mixer.phase = 0;
    for (uint32_t i = 0; i < c->fft_size; ++i) d->channel_estimate[i] = c_make(1, 0);
    d->snr_symbol_count = 0; d->estimated_snr_linear = 1.0f; d->noise_variance = 0.1f;
    d->symbols_since_sync = 0; d->n_prev = 0; d->pilot_phase_correction = c_make(1, 0);
    d->n_dprev = 0; d->carrier_phase_initialized = 0; d->carrier_phase_correction = c_make(1, 0);
    adaptive_eq_reset(d);                                   /* :894-897 */
    const float* ptr = audio; size_t remaining = n_samples;
    if (c->training_symbols > 0) {
        size_t tcount = (size_t)c->training_symbols * d->symbol_samples;
        estimate_channel_from_lts(d, ptr, c->training_symbols);
        ptr += tcount; remaining -= tcount;
    }
    d->n_dprev = 0;
    while (remaining >= d->symbol_samples) {
        demod_symbol(d, ptr, bb, freq, eq, d->cr.n_pilot != 0);
        ptr += d->symbol_samples; remaining -= d->symbol_samples;
    }
    return 1;
}

int uo_demod_presynced(const ultra_hip_config* c, const float* audio, uint32_t n_samples,
                       int has_cfo, float cfo_hz, float cfo_phase,
                       float* llr_out, uint32_t llr_cap, float* H_out, float* scal_out) {
    if (!has_cfo) cfo_hz = NAN;              /* never set: the training-symbol estimate (ofdm_sync.cpp:278-380) */
    demod* d = (demod*)malloc(sizeof(demod));
    if (demod_init(d, c) != 0) { free(d); return -1; }
    d->soft = llr_out; d->soft_cap = llr_cap; d->n_soft = 0;
    cf* bb = (cf*)malloc(sizeof(cf) * d->symbol_samples);
    cf freq[MAX_FFT], eq[MAX_CARRIERS];
    demod_presynced_run(d, audio, n_samples, cfo_hz, cfo_phase, bb, freq, eq);
    if (H_out) memcpy(H_out, d->channel_estimate, sizeof(cf) * c->fft_size);
    if (scal_out) dump_scal(d, scal_out);
    int n = (int)d->n_soft;
    free(bb); demod_free(d); free(d);
    return n;
}

int uo_demod_tables(const ultra_hip_config* c, int32_t* data_idx, int32_t* pilot_idx,
                    float* pilot_seq_ri, int32_t* interp_i, float* interp_alpha,
                    float* sync_seq_ri, uint32_t* counts) {
    demod* d = (demod*)malloc(sizeof(demod));
    if (demod_init(d, c) != 0) { free(d); return -1; }
    counts[0] = (uint32_t)d->cr.n_data; counts[1] = (uint32_t)d->cr.n_pilot;
    counts[2] = (uint32_t)d->n_interp; counts[3] = (uint32_t)d->cr.n_sync;
    for (int i = 0; i < d->cr.n_data; ++i) data_idx[i] = d->cr.data_idx[i];
    for (int i = 0; i < d->cr.n_pilot; ++i) {
        pilot_idx[i] = d->cr.pilot_idx[i];
        pilot_seq_ri[2 * i] = d->cr.pilot_seq[i].re; pilot_seq_ri[2 * i + 1] = d->cr.pilot_seq[i].im;
    }
    for (int i = 0; i < d->n_interp; ++i) {
        interp_i[3 * i] = d->interp[i].fft_idx; interp_i[3 * i + 1] = d->interp[i].lower_pilot;
        interp_i[3 * i + 2] = d->interp[i].upper_pilot; interp_alpha[i] = d->interp[i].alpha;
    }
    for (int i = 0; i < d->cr.n_sync; ++i) {
        sync_seq_ri[2 * i] = d->cr.sync_seq[i].re; sync_seq_ri[2 * i + 1] = d->cr.sync_seq[i].im;
    }
    demod_free(d); free(d);
    return 0;
}

/* ---------------------------------------------------------------------- */
/* batched receive path (the CPU twin of libultra_hip.so)                  */
/* ---------------------------------------------------------------------- */
typedef struct batch_job {
    const ultra_hip_config* c; const float* audio; size_t stride;
    const float* cfo_hz; const float* cfo_phase; size_t f0, f1;
    float* llr_out; float* state_out; uint8_t* bytes_out; int32_t* iters_out; uint8_t* ok_out;
    int rc;
} batch_job;

static void* batch_worker(void* arg) {
    batch_job* j = (batch_job*)arg;
    const ultra_hip_config* c = j->c;
    ultra_hip_geometry g;
    if (uo_geometry(c, &g) != 0) { j->rc = -1; return NULL; }
    demod* d = (demod*)malloc(sizeof(demod));
    if (demod_init(d, c) != 0) { free(d); j->rc = -1; return NULL; }
    const ldpc_code* code = ldpc_get(c->code_rate);
    size_t S = d->symbol_samples;
    cf* bb = (cf*)malloc(sizeof(cf) * S);
    cf freq[MAX_FFT], eq[MAX_CARRIERS];
    float* llr = (float*)malloc(sizeof(float) * (g.llrs_per_frame + 8));
    for (size_t f = j->f0; f < j->f1; ++f) {
        const float* a = j->audio + f * j->stride;
        float cfo = j->cfo_hz ? j->cfo_hz[f] : 0.0f;
        float ph = j->cfo_phase ? j->cfo_phase[f] : 0.0f;
        /* a fresh demodulator per frame, as every harness constructs one
         * (tools/test_otfs_vs_ofdm.cpp:118-120): reset all per-frame state */
        for (uint32_t i = 0; i < c->fft_size; ++i) d->channel_estimate[i] = c_make(1, 0);
        d->noise_variance = 0.1f; d->estimated_snr_linear = 1.0f; d->snr_symbol_count = 0;
        d->n_prev = 0; d->pilot_phase_correction = c_make(1, 0); d->timing_offset_samples = 0.0f;
        d->n_lts = 0; d->n_dprev = 0; d->mixer.phase = 0;
        adaptive_eq_reset(d);
        d->soft = llr; d->soft_cap = g.llrs_per_frame; d->n_soft = 0;
        if (c->entry == ULTRA_ENTRY_PRESYNCED) {
            demod_presynced_run(d, a, g.frame_samples, cfo, ph, bb, freq, eq);
        } else {
            demod_enter_synced(d, cfo);
            d->freq_correction_phase = ph;
            for (uint32_t s = 0; s < c->n_data_symbols; ++s) demod_symbol(d, a + (size_t)s * S, bb, freq, eq, 1);
        }
        if (j->llr_out) memcpy(j->llr_out + f * g.llrs_per_frame, llr, sizeof(float) * g.llrs_per_frame);
        if (j->state_out) {
            float* st = j->state_out + f * ULTRA_HIP_STATE_FLOATS;
            st[ULTRA_HIP_STATE_FREQ_OFFSET_HZ] = d->freq_offset_hz;
            st[ULTRA_HIP_STATE_NOISE_VARIANCE] = d->noise_variance;
            st[ULTRA_HIP_STATE_SNR_LINEAR] = d->estimated_snr_linear;
            st[ULTRA_HIP_STATE_TIMING_OFFSET] = d->timing_offset_samples;
            st[ULTRA_HIP_STATE_CFO_PHASE] = d->freq_correction_phase;
            st[ULTRA_HIP_STATE_MIXER_PHASE] = d->mixer.phase;
            st[ULTRA_HIP_STATE_SYMBOLS] = (float)d->snr_symbol_count;
            st[ULTRA_HIP_STATE_RESERVED] = 0.0f;
        }
        if (j->bytes_out) {
            /* first 648 soft bits → decoder (tools/test_nvis_mode.cpp:96-103);
             * fewer than 648 available = frame failure there (returns false) */
            float total[LDPC_N]; int ok = 0, it = 0;
            uint8_t bits[LDPC_N];
            uint8_t* ob = j->bytes_out + f * g.decoded_bytes;
            if (g.llrs_per_frame >= LDPC_N) {
                ldpc_decode_block(code, (int)c->max_iterations, llr, LDPC_N, total, &ok, &it);
                for (int q = 0; q < code->k; ++q) bits[q] = (total[q] < 0) ? 1 : 0;
                pack_bits(bits, code->k, ob, g.decoded_bytes);
            } else {
                memset(ob, 0, g.decoded_bytes); ok = 0; it = 0;
            }
            j->iters_out[f] = it; j->ok_out[f] = (uint8_t)ok;
        }
    }
    free(llr); free(bb); demod_free(d); free(d);
    j->rc = 0;
    return NULL;
}

int uo_demod_decode_batch(const ultra_hip_config* c, const float* audio, size_t frame_stride,
                          const float* cfo_hz, const float* cfo_phase, size_t n_frames,
                          int n_threads, float* llr_out, float* state_out,
                          uint8_t* bytes_out, int32_t* iters_out, uint8_t* ok_out) {
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_frames) n_threads = n_frames ? (int)n_frames : 1;
    (void)ldpc_get(c->code_rate);
    batch_job* jobs = (batch_job*)calloc((size_t)n_threads, sizeof(batch_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; ++t) {
        batch_job* j = &jobs[t];
        j->c = c; j->audio = audio; j->stride = frame_stride; j->cfo_hz = cfo_hz; j->cfo_phase = cfo_phase;
        j->f0 = n_frames * (size_t)t / (size_t)n_threads; j->f1 = n_frames * (size_t)(t + 1) / (size_t)n_threads;
        j->llr_out = llr_out; j->state_out = state_out; j->bytes_out = bytes_out; j->iters_out = iters_out; j->ok_out = ok_out;
        if (n_threads == 1) batch_worker(j); else pthread_create(&th[t], NULL, batch_worker, j);
    }
    int rc = 0;
    for (int t = 0; t < n_threads; ++t) { if (n_threads > 1) pthread_join(th[t], NULL); if (jobs[t].rc) rc = jobs[t].rc; }
    free(jobs); free(th);
    return rc;
}

/* ====================================================================== */
/* Acquisition: SEARCHING state of OFDMDemodulator::process (scope row f1) */
/* src/ofdm/demodulator.cpp:461-600, src/ofdm/ofdm_sync.cpp:20-261,386-461 */
/* ====================================================================== */
#define ACQ_MIN_SEARCH_SAMPLES 4000u      /* demodulator_constants.hpp:41-53 */
#define ACQ_MAX_BUFFER_SAMPLES 240000u
#define ACQ_OVERLAP_SAMPLES 20000u
#define ACQ_SEARCH_STEP 8u
#define ACQ_PLATEAU_WINDOW 300u
#define ACQ_MIN_PLATEAU 15u
#define ACQ_LTS_MAX (MAX_FFT + 256)

typedef struct acq {
    demod d;
    float sync_threshold;          /* ModemConfig::sync_threshold = 0.80f (types.hpp:188) */
    float noise_floor_energy;      /* demodulator_impl.hpp:62 */
    size_t lts_len;
    float lts_I[ACQ_LTS_MAX], lts_Q[ACQ_LTS_MAX];
    cf work[MAX_FFT];
    float dc_removed[MAX_FFT];
} acq;

/* LTS passband templates of the constructor, src/ofdm/demodulator.cpp:100-133 */
static int acq_init(acq* a, const ultra_hip_config* c) {
    if (demod_init(&a->d, c) != 0) return -1;
    a->sync_threshold = (c->sync_threshold != 0.0f) ? c->sync_threshold : 0.80f;
    a->noise_floor_energy = 0.0f;
    const uint32_t N = c->fft_size, cp = a->d.cp;
    if (cp + N > ACQ_LTS_MAX) return -1;
    cf lts[MAX_FFT];
    for (uint32_t i = 0; i < N; ++i) lts[i] = c_make(0, 0);
    for (int i = 0; i < a->d.cr.n_data; ++i) lts[a->d.cr.data_idx[i]] = a->d.cr.sync_seq[i % a->d.cr.n_sync];
    for (int i = 0; i < a->d.cr.n_pilot; ++i) lts[a->d.cr.pilot_idx[i]] = a->d.cr.pilot_seq[i];
    fft_exec(&a->d.fft, lts, 1);
    nco o; nco_init(&o, (float)c->center_freq, (float)c->sample_rate);
    a->lts_len = cp + N;
    for (uint32_t i = 0; i < cp + N; ++i) {
        cf base = (i < cp) ? lts[N - cp + i] : lts[i - cp];
        cf mixed = c_mul(base, nco_next(&o));
        a->lts_I[i] = mixed.re;
        a->lts_Q[i] = mixed.im;
    }
    return 0;
}

/* Impl::hasMinimumEnergy, ofdm_sync.cpp:20-50 (stateful: noise floor) */
static int acq_has_energy(acq* a, const float* buf, size_t size, size_t offset, size_t window_len) {
    if (offset + window_len > size) return 0;
    float sum_sq = 0; size_t count = 0;
    for (size_t i = 0; i < window_len; i += 16) { float s = buf[offset + i]; sum_sq += s * s; ++count; }
    float energy = sum_sq / (float)count;
    if (a->noise_floor_energy < 1e-20f) a->noise_floor_energy = energy * 0.1f;
    if (energy < a->noise_floor_energy) a->noise_floor_energy = energy;
    else if (energy < a->noise_floor_energy * 3.0f)
        a->noise_floor_energy = (1.0f - 0.01f) * a->noise_floor_energy + 0.01f * energy;
    float threshold = a->noise_floor_energy * 4.0f;
    return energy >= threshold;
}

/* Impl::toAnalytic, ofdm_sync.cpp:56-84, for len == fft_size (a power of two: no padding) */
static void acq_analytic(acq* a, const float* samples, cf* out) {
    const uint32_t N = a->d.cfg.fft_size;
    for (uint32_t i = 0; i < N; ++i) out[i] = c_make(samples[i], 0);
    fft_exec(&a->d.fft, out, 0);
    for (uint32_t i = 1; i < N / 2; ++i) out[i] = c_scale(out[i], 2.0f);
    for (uint32_t i = N / 2 + 1; i < N; ++i) out[i] = c_make(0, 0);
    fft_exec(&a->d.fft, out, 1);
}

/* Impl::measureSchmidlCoxCorrelation, ofdm_sync.cpp:120-163 */
static float acq_sc(acq* a, const float* buf, size_t size, size_t offset, cf* out_P, float* out_energy) {
    const size_t cp = a->d.cp, N = a->d.cfg.fft_size, half = N / 2;
    if (offset + cp + N > size) { if (out_energy) *out_energy = 0.0f; return 0.0f; }
    const size_t ds = offset + cp;
    float dc_sum = 0.0f;
    for (size_t i = 0; i < N; ++i) dc_sum += buf[ds + i];
    float dc_offset = dc_sum / (float)N;
    for (size_t i = 0; i < N; ++i) a->dc_removed[i] = buf[ds + i] - dc_offset;
    acq_analytic(a, a->dc_removed, a->work);
    cf P = c_make(0, 0); float R1 = 0, R2 = 0;
    for (size_t i = 0; i < half; ++i) {
        P = c_add(P, c_mul(c_conj(a->work[i]), a->work[i + half]));
        R1 += c_norm(a->work[i]);
        R2 += c_norm(a->work[i + half]);
    }
    if (out_P) *out_P = P;
    if (out_energy) *out_energy = R2;
    float normalization = sqrtf(R1 * R2);
    if (normalization < 1e-10f) return 0.0f;
    return c_abs(P) / normalization;
}

/* Impl::estimateCoarseCFO, ofdm_sync.cpp:230-261 */
static float acq_coarse_cfo(acq* a, const float* buf, size_t size, size_t sync_offset) {
    const size_t cp = a->d.cp, N = a->d.cfg.fft_size, half = N / 2;
    const size_t ds = sync_offset + cp;
    if (ds + N > size) return 0.0f;
    acq_analytic(a, buf + ds, a->work);
    cf P = c_make(0, 0);
    for (size_t i = 0; i < half; ++i) P = c_add(P, c_mul(c_conj(a->work[i]), a->work[i + half]));
    float phase = atan2f(P.im, P.re);
    /* float cfo_hz = phase * config.sample_rate / (M_PI * fft_len);  float*uint32 -> float, / double */
    float cfo_hz = (float)((double)(phase * (float)a->d.cfg.sample_rate) / (M_PI * (double)N));
    /* float max_cfo = config.sample_rate / fft_len;  integer division */
    float max_cfo = (float)((size_t)a->d.cfg.sample_rate / N);
    return f_max(-max_cfo, f_min(max_cfo, cfo_hz));
}

/* Impl::refineLTSTiming, ofdm_sync.cpp:386-461; SIZE_MAX = failure */
static size_t acq_refine_lts(acq* a, const float* buf, size_t size, size_t coarse_sts_start) {
    const size_t psl = a->d.cfg.fft_size + a->d.cp;
    const size_t coarse_lts_start = coarse_sts_start + 4 * psl;
    const int SEARCH_BACK = (int)(3 * psl), SEARCH_FWD = (int)(psl / 2);
    if (coarse_lts_start < (size_t)SEARCH_BACK || coarse_lts_start + (size_t)SEARCH_FWD + a->lts_len > size)
        return coarse_lts_start;
    float best_corr = 0.0f; size_t best_offset = coarse_lts_start;
    float energy_ref = 0.0f;
    for (size_t i = 0; i < a->lts_len; ++i) {
        energy_ref += a->lts_I[i] * a->lts_I[i];
        energy_ref += a->lts_Q[i] * a->lts_Q[i];
    }
    energy_ref *= 0.5f;
    for (int delta = -SEARCH_BACK; delta <= SEARCH_FWD; ++delta) {
        size_t offset = coarse_lts_start + (size_t)(long)delta;
        float corr_I = 0, corr_Q = 0, energy_rx = 0;
        for (size_t i = 0; i < a->lts_len; ++i) {
            float rx = buf[offset + i];
            corr_I += rx * a->lts_I[i];
            corr_Q += rx * a->lts_Q[i];
            energy_rx += rx * rx;
        }
        float corr_mag = sqrtf(corr_I * corr_I + corr_Q * corr_Q);
        float norm = sqrtf(energy_rx * energy_ref);
        float corr = (norm > 1e-6f) ? corr_mag / norm : 0.0f;
        if (corr > best_corr) { best_corr = corr; best_offset = offset; }
    }
    float thr = (a->d.cfg.fft_size >= 1024) ? 0.05f : 0.35f;
    if (best_corr < thr) return (size_t)-1;
    return best_offset;
}

/* One stream fed in `chunk`-sample calls until sync is declared.  data_start is absolute in the
 * stream (process() erases the buffer up to it: demodulator.cpp:572-575). */
int uo_acquire(const ultra_hip_config* c, const float* audio, uint32_t n, uint32_t chunk,
               uint32_t* found, uint32_t* fed_at_sync, uint32_t* sync_offset, float* coarse_cfo,
               uint32_t* refined_lts, uint32_t* data_start, float* noise_floor) {
    acq* a = (acq*)malloc(sizeof(acq));
    if (!a) return -1;
    if (chunk == 0 || acq_init(a, c) != 0) { free(a); return -1; }
    *found = 0; *fed_at_sync = 0; *sync_offset = 0; *coarse_cfo = 0; *refined_lts = 0; *data_start = 0;
    const size_t psl = c->fft_size + a->d.cp, preamble_total = psl * 6, corr_win = psl * 2;
    size_t base = 0, fed = 0;
    while (fed < n && !*found) {
        fed += (n - fed < chunk) ? (n - fed) : chunk;
        size_t size = fed - base;
        if (size < ACQ_MIN_SEARCH_SAMPLES) continue;
        if (size > ACQ_MAX_BUFFER_SAMPLES) { base = fed - ACQ_OVERLAP_SAMPLES; size = ACQ_OVERLAP_SAMPLES; }
        const float* buf = audio + base;
        int found_sync = 0; size_t so = 0;
        size_t search_end = (size > preamble_total + corr_win) ? size - preamble_total - corr_win : 0;
        for (size_t i = 0; i < search_end; i += ACQ_SEARCH_STEP) {
            if (!acq_has_energy(a, buf, size, i, corr_win)) { i += corr_win / 2 - ACQ_SEARCH_STEP; continue; }
            float corr = acq_sc(a, buf, size, i, NULL, NULL);
            if (corr > a->sync_threshold) {
                size_t plateau = 0, peak_pos = i; float peak = corr;
                for (size_t j = 0; j <= ACQ_PLATEAU_WINDOW && i + j + preamble_total < size; j += 8) {
                    float rc = acq_sc(a, buf, size, i + j, NULL, NULL);
                    if (rc >= 0.90f) plateau++;
                    if (rc > peak) { peak = rc; peak_pos = i + j; }
                }
                if (plateau >= ACQ_MIN_PLATEAU) { found_sync = 1; so = peak_pos; break; }
            }
        }
        if (found_sync) {
            float cfo = acq_coarse_cfo(a, buf, size, so);
            size_t refined = acq_refine_lts(a, buf, size, so);
            if (refined == (size_t)-1) {
                if (size > ACQ_OVERLAP_SAMPLES * 2) {
                    size_t trim = so + psl;
                    if (trim > size - ACQ_OVERLAP_SAMPLES) trim = size - ACQ_OVERLAP_SAMPLES;
                    base += trim;
                }
            } else {
                *found = 1; *fed_at_sync = (uint32_t)fed; *sync_offset = (uint32_t)so; *coarse_cfo = cfo;
                *refined_lts = (uint32_t)refined;
                *data_start = (uint32_t)(base + refined + 2 * psl);
            }
        } else if (size > ACQ_OVERLAP_SAMPLES * 2) {
            base += size - ACQ_OVERLAP_SAMPLES;
        }
    }
    *noise_floor = a->noise_floor_energy;
    demod_free(&a->d);
    free(a);
    return 0;
}

/* The preamble check of the SYNCED state, src/ofdm/demodulator.cpp:605-657, on rx_buffer = audio[0, n): the first offset
 * 0, 8, .. <= min(n - 6 preamble symbols, 2 data symbols) whose Schmidl-Cox metric exceeds the threshold AND whose LTS
 * confirmation holds (a failed confirmation continues the scan).  consume = what process() erases (:629).  The counters
 * that arm the check (synced_symbol_count > 0, idle_call_count >= 2) are the caller's. */
int uo_midframe_search(const ultra_hip_config* c, const float* audio, uint32_t n, uint32_t* found, uint32_t* sts_start,
                       uint32_t* refined_lts, uint32_t* consume, float* coarse_cfo) {
    acq* a = (acq*)malloc(sizeof(acq));
    if (!a) return -1;
    if (acq_init(a, c) != 0) { free(a); return -1; }
    *found = 0; *sts_start = 0; *refined_lts = 0; *consume = 0; *coarse_cfo = 0;
    const size_t psl = c->fft_size + a->d.cp, preamble_total = psl * 6;
    const size_t symbol_samples = c->fft_size + a->d.cp + c->symbol_guard;
    if (n >= preamble_total) {
        size_t search_limit = n - preamble_total;
        if (symbol_samples * 2 < search_limit) search_limit = symbol_samples * 2;
        for (size_t offset = 0; offset <= search_limit; offset += 8) {
            float corr = acq_sc(a, audio, n, offset, NULL, NULL);
            if (corr > a->sync_threshold) {
                size_t refined = acq_refine_lts(a, audio, n, offset);
                if (refined == (size_t)-1) continue;
                *found = 1; *sts_start = (uint32_t)offset; *refined_lts = (uint32_t)refined;
                *consume = (uint32_t)(refined + 2 * psl);
                *coarse_cfo = acq_coarse_cfo(a, audio, n, offset);
                break;
            }
        }
    }
    demod_free(&a->d);
    free(a);
    return 0;
}

/* Stage probes for the tests (same outputs as ref_sc_metric / ref_lts_templates in oracle/ref_shim.cpp) */
int uo_sc_metric(const ultra_hip_config* c, const float* audio, uint32_t n, uint32_t offset,
                 float* corr, float* p_re, float* p_im, float* energy, float* noise_floor_io, uint32_t* has_energy) {
    acq* a = (acq*)malloc(sizeof(acq));
    if (!a) return -1;
    if (acq_init(a, c) != 0) { free(a); return -1; }
    cf P = c_make(0, 0); float e = 0;
    *corr = acq_sc(a, audio, n, offset, &P, &e);
    *p_re = P.re; *p_im = P.im; *energy = e;
    a->noise_floor_energy = *noise_floor_io;
    *has_energy = (uint32_t)acq_has_energy(a, audio, n, offset, 2 * (size_t)(c->fft_size + a->d.cp));
    *noise_floor_io = a->noise_floor_energy;
    demod_free(&a->d); free(a);
    return 0;
}
int uo_lts_templates(const ultra_hip_config* c, float* I, float* Q, uint32_t cap) {
    acq* a = (acq*)malloc(sizeof(acq));
    if (!a) return -1;
    if (acq_init(a, c) != 0) { free(a); return -1; }
    int m = (int)a->lts_len;
    if ((uint32_t)m > cap) { demod_free(&a->d); free(a); return -1; }
    memcpy(I, a->lts_I, sizeof(float) * m); memcpy(Q, a->lts_Q, sizeof(float) * m);
    demod_free(&a->d); free(a);
    return m;
}

/* ====================================================================== */
/* Chirp synchronisation (scope row f4): sync::ChirpSync, src/sync/chirp_sync.hpp,  */
/* as configured by OFDMChirpWaveform (src/waveform/ofdm_chirp_waveform.cpp:39-49,   */
/* 129-172): 300 -> 2700 Hz up chirp, 100 ms gap, down chirp, 500 ms each.          */
/* ====================================================================== */
typedef struct chirp_sync {
    float fs, f_start, f_end, duration_ms, gap_ms, amplitude, tx_cfo;
    size_t len;
    float *up_s, *up_c, *dn_s, *dn_c;
    float e_up, e_dn;
} chirp_sync;

static float chirp_up_phase(const chirp_sync* c, float t) {      /* chirp_sync.hpp:687-691 */
    float T = c->duration_ms / 1000.0f;
    float k = (c->f_end - c->f_start) / T;
    return (float)(2.0f * M_PI * (double)(c->f_start * t + 0.5f * k * t * t));
}
static float chirp_down_phase(const chirp_sync* c, float t) {    /* :694-699 */
    float T = c->duration_ms / 1000.0f;
    float k = (c->f_end - c->f_start) / T;
    return (float)(2.0f * M_PI * (double)(c->f_end * t - 0.5f * k * t * t));
}
static int chirp_init(chirp_sync* c, float sample_rate, float tx_cfo) {   /* ctor + generateTemplate :706-732 */
    c->fs = sample_rate; c->f_start = 300.0f; c->f_end = 2700.0f; c->duration_ms = 500.0f; c->gap_ms = 100.0f;
    c->amplitude = 0.5f; c->tx_cfo = tx_cfo;
    c->len = (size_t)(c->fs * c->duration_ms / 1000.0f);
    c->up_s = (float*)malloc(sizeof(float) * c->len * 4);
    if (!c->up_s) return -1;
    c->up_c = c->up_s + c->len; c->dn_s = c->up_c + c->len; c->dn_c = c->dn_s + c->len;
    c->e_up = 0.0f; c->e_dn = 0.0f;
    for (size_t i = 0; i < c->len; ++i) {
        float t = (float)i / c->fs;
        float ph = chirp_up_phase(c, t);
        c->up_s[i] = sinf(ph); c->up_c[i] = cosf(ph);
        c->e_up += c->up_s[i] * c->up_s[i];
    }
    for (size_t i = 0; i < c->len; ++i) {
        float t = (float)i / c->fs;
        float ph = chirp_down_phase(c, t);
        c->dn_s[i] = sinf(ph); c->dn_c[i] = cosf(ph);
        c->e_dn += c->dn_s[i] * c->dn_s[i];
    }
    return 0;
}
static void chirp_free(chirp_sync* c) { free(c->up_s); c->up_s = NULL; }

/* computeComplexTemplateCorrelation, chirp_sync.hpp:532-556 */
static float chirp_corr(const float* x, size_t n, size_t offset, const float* ts, const float* tc, size_t len, float energy) {
    if (offset + len > n) return 0.0f;
    float cI = 0.0f, cQ = 0.0f, se = 0.0f;
    for (size_t i = 0; i < len; ++i) {
        float s = x[offset + i];
        cI += s * tc[i];
        cQ += s * ts[i];
        se += s * s;
    }
    float denom = sqrtf(se * energy);
    if (denom < 1e-10f) return 0.0f;
    return sqrtf(cI * cI + cQ * cQ) / denom;
}

/* detectChirpTemplate, chirp_sync.hpp:560-632 */
static int chirp_detect_template(const float* x, size_t n, const float* ts, const float* tc, size_t len, float energy,
                                 float threshold, float* corr_out) {
    if (n < len) { *corr_out = 0.0f; return -1; }
    const size_t search_len = n - len;
    float best_corr = 0.0f; int best_pos = -1;
    for (size_t pos = 0; pos < search_len; pos += 48) {
        float c = chirp_corr(x, n, pos, ts, tc, len, energy);
        if (c > best_corr) { best_corr = c; best_pos = (int)pos; }
    }
    if (best_pos < 0 || best_corr < threshold * 0.3f) { *corr_out = best_corr; return -1; }
    int fine_start = best_pos - 48; if (fine_start < 0) fine_start = 0;
    int fine_end = best_pos + 48; if (fine_end > (int)search_len) fine_end = (int)search_len;
    for (int pos = fine_start; pos <= fine_end; ++pos) {
        float c = chirp_corr(x, n, (size_t)pos, ts, tc, len, energy);
        if (c > best_corr) { best_corr = c; best_pos = pos; }
    }
    if (best_pos > 0 && best_pos < (int)search_len - 1) {
        float c0 = chirp_corr(x, n, (size_t)(best_pos - 1), ts, tc, len, energy);
        float c1 = best_corr;
        float c2 = chirp_corr(x, n, (size_t)(best_pos + 1), ts, tc, len, energy);
        float denom = 2.0f * (c0 - 2.0f * c1 + c2);
        if (fabsf(denom) > 1e-10f) {
            float delta = (c0 - c2) / denom;
            delta = f_max(-1.0f, f_min(1.0f, delta));
            best_pos = (int)roundf((float)best_pos + delta);
        }
    }
    *corr_out = best_corr;
    return (best_corr >= threshold) ? best_pos : -1;
}

/* detectDualChirp (:349-505) + OFDMChirpWaveform::detectSync's start_sample (ofdm_chirp_waveform.cpp:129-172).
 * out[0..5] = success, up_chirp_start, down_chirp_start, start_sample (training start), raw up_pos, raw down_pos;
 * fout[0..2] = cfo_hz, up_correlation, down_correlation */
int uo_chirp_detect(float sample_rate, const float* x, uint32_t n, float threshold, int32_t* out, float* fout) {
    chirp_sync c;
    if (chirp_init(&c, sample_rate, 0.0f) != 0) return -1;
    for (int i = 0; i < 6; ++i) out[i] = (i == 0) ? 0 : -1;
    fout[0] = fout[1] = fout[2] = 0.0f;
    const size_t len = c.len, gap = (size_t)(c.fs * c.gap_ms / 1000.0f);
    if ((size_t)n >= 2 * len + gap) {
        float up_corr = 0.0f;
        int up_pos = chirp_detect_template(x, n, c.up_s, c.up_c, len, c.e_up, threshold, &up_corr);
        fout[1] = up_corr;
        if (up_pos >= 0) {
            out[4] = up_pos;
            size_t ds = (size_t)up_pos + len / 2;
            size_t expected_down = (size_t)up_pos + len + gap;
            size_t de = expected_down + 2 * len; if (de > n) de = n;
            if (ds < n) {
                if (de <= ds + len) { de = ds + 2 * len; if (de > n) de = n; }
                float dn_corr = 0.0f;
                int dn_rel = chirp_detect_template(x + ds, de - ds, c.dn_s, c.dn_c, len, c.e_dn, threshold, &dn_corr);
                if (dn_rel >= 0) {
                    int dn_pos = dn_rel + (int)ds;
                    out[5] = dn_pos; fout[2] = dn_corr;
                    float T = c.duration_ms / 1000.0f;
                    float chirp_rate = (c.f_end - c.f_start) / T;
                    float cfo_to_samples = c.fs / chirp_rate;
                    int expected_gap = (int)(len + gap);
                    int actual_gap = dn_pos - up_pos;
                    float gap_error = (float)(actual_gap - expected_gap);
                    float cfo = gap_error / (2.0f * cfo_to_samples);
                    fout[0] = cfo;
                    if (!(fabsf(cfo) > 100.0f)) {
                        float up_correction = cfo * cfo_to_samples, down_correction = -cfo * cfo_to_samples;
                        out[1] = (int)roundf((float)up_pos + up_correction);
                        out[2] = (int)roundf((float)dn_pos + down_correction);
                        out[0] = 1;
                        /* result.start_sample = down_chirp_start + chirp_samples + gap_samples */
                        size_t gap2 = (size_t)((float)(uint32_t)sample_rate * 100.0f / 1000.0f);
                        out[3] = (int)((size_t)out[2] + len + gap2);
                    }
                }
            }
        }
    }
    chirp_free(&c);
    return 0;
}

/* ChirpSync::generate (:37-106): up chirp, gap, down chirp, gap */
int uo_chirp_generate(float sample_rate, float tx_cfo_hz, float* out, uint32_t cap) {
    chirp_sync c;
    if (chirp_init(&c, sample_rate, tx_cfo_hz) != 0) return -1;
    size_t len = c.len, gap = (size_t)(c.fs * c.gap_ms / 1000.0f), total = 2 * len + 2 * gap;
    if (total > cap) { chirp_free(&c); return -1; }
    for (size_t i = 0; i < total; ++i) out[i] = 0.0f;
    float T = c.duration_ms / 1000.0f, k = (c.f_end - c.f_start) / T;
    float f_up = c.f_start + c.tx_cfo, f_dn = c.f_end + c.tx_cfo;
    for (size_t i = 0; i < len; ++i) {
        float t = (float)i / c.fs;
        float ph = (float)(2.0f * M_PI * (double)(f_up * t + 0.5f * k * t * t));
        out[i] = c.amplitude * sinf(ph);
    }
    for (size_t i = 0; i < len; ++i) {
        float t = (float)i / c.fs;
        float ph = (float)(2.0f * M_PI * (double)(f_dn * t - 0.5f * k * t * t));
        out[len + gap + i] = c.amplitude * sinf(ph);
    }
    chirp_free(&c);
    return (int)total;
}

int uo_chirp_templates(float sample_rate, float* up_s, float* up_c, float* dn_s, float* dn_c, float* energies, uint32_t cap) {
    chirp_sync c;
    if (chirp_init(&c, sample_rate, 0.0f) != 0) return -1;
    if (c.len > cap) { chirp_free(&c); return -1; }
    memcpy(up_s, c.up_s, 4 * c.len); memcpy(up_c, c.up_c, 4 * c.len);
    memcpy(dn_s, c.dn_s, 4 * c.len); memcpy(dn_c, c.dn_c, 4 * c.len);
    energies[0] = c.e_up; energies[1] = c.e_dn;
    int m = (int)c.len;
    chirp_free(&c);
    return m;
}

/* ====================================================================== */
/* v2 wire format on the receive path (scope row f4, second half):         */
/* RxPipeline::processFrame from the soft bits on — detectPing,            */
/* deinterleaveCodewords, decodeFrame (src/gui/modem/rx_pipeline.cpp:       */
/* 283-346,348-444,446-511) with v2::parseHeader / identifyCodeword /      */
/* CodewordStatus::reassemble (src/protocol/frame_v2.cpp:952-982,          */
/* 1023-1044,1175-1229) — and the transmit side that makes the stimulus    */
/* (DataFrame::serialize :502-554, ControlFrame::serialize :346-382,       */
/* encodeFrameWithLDPC :1079-1128).                                        */
/* ====================================================================== */
/* ControlFrame::calculateCRC, frame_v2.cpp:111-124 (CRC-16/CCITT-FALSE) */
uint16_t uo_crc16(const uint8_t* d, uint32_t n) {
    uint16_t crc = 0xFFFF;
    for (uint32_t i = 0; i < n; ++i) {
        crc ^= (uint16_t)((uint16_t)d[i] << 8);
        for (int j = 0; j < 8; ++j) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ 0x1021) : (uint16_t)(crc << 1);
    }
    return crc;
}

static int v2_is_control(uint8_t t) {   /* isControlFrame, frame_v2.hpp:212-217 */
    return t == 0x10 || t == 0x11 || t == 0x16 || t == 0x17 || t == 0x20 || t == 0x21 || t == 0x40;
}

/* v2::parseHeader, frame_v2.cpp:1175-1229.  out = {type, total_cw, payload_len, is_control}; returns valid */
int uo_v2_parse_header(const uint8_t* d, uint32_t n, int32_t* out) {
    out[0] = 0x10; out[1] = 0; out[2] = 0; out[3] = 0;          /* HeaderInfo defaults */
    if (n < 20) return 0;
    if (d[0] != 0x55 || d[1] != 0x4C) return 0;
    out[0] = d[2];
    out[3] = v2_is_control(d[2]);
    if (out[3]) {
        uint16_t rx = (uint16_t)((d[18] << 8) | d[19]);
        if (rx != uo_crc16(d, 18)) return 0;
        out[1] = 1; out[2] = 0;
    } else {
        out[1] = d[12];
        out[2] = (d[13] << 8) | d[14];
        uint16_t rx = (uint16_t)((d[15] << 8) | d[16]);
        if (rx != uo_crc16(d, 15)) return 0;
    }
    return 1;
}

static const int v2_info_bits[6] = {162, 216, 324, 432, 486, 540};    /* getInfoBitsForRate, frame_v2.hpp:551-561 */

/* res = {success, is_ping, frame_type, codewords_ok, codewords_failed, expected_codewords (accumulating: CW0's
 * total_cw when fewer codewords were handed in, else 0), frame_len, status}; status: 0 no codeword / CW0 failed,
 * 1 invalid header, 2 waiting for more codewords, 3 some codeword failed, 4 frame complete, 5 ping.
 * deint_bps: RxPipeline::setInterleaverConfig's bits_per_symbol with interleaving enabled, 0 = disabled. */
int uo_v2_decode_frame(uint32_t rate, uint32_t deint_bps, int max_iters, const float* soft, uint32_t n_soft,
                       int32_t* res, uint8_t* frame_data, uint32_t cap) {
    for (int i = 0; i < 8; ++i) res[i] = 0;
    res[2] = 0x10;                                               /* RxFrameResult::frame_type = PROBE */
    if (rate > 5) return -1;
    if (n_soft == 0) return 0;                                   /* "No soft bits from waveform" */
    /* detectPing (:446-472): hard bytes with soft > 0 -> bit 1, "ULTR" or its inversion */
    {
        uint8_t pb[8]; uint32_t np = 0;
        for (uint32_t i = 0; i + 8 <= n_soft && np < 8; i += 8) {
            uint8_t byte = 0;
            for (int b = 0; b < 8; ++b) if (soft[i + b] > 0) byte |= (uint8_t)(1 << (7 - b));
            pb[np++] = byte;
        }
        if (np >= 4) {
            int normal = pb[0] == 0x55 && pb[1] == 0x4C && pb[2] == 0x54 && pb[3] == 0x52;
            int inverted = pb[0] == 0xAA && pb[1] == 0xB3 && pb[2] == 0xAB && pb[3] == 0xAD;
            if (normal || inverted) { res[0] = 1; res[1] = 1; res[2] = 0x01; res[7] = 5; return 0; }
        }
    }
    const uint32_t num_cw = n_soft / 648;
    if (num_cw == 0) return 0;
    uint32_t perm[648], inv[648];
    if (deint_bps) uo_channel_interleaver_perm(deint_bps, 648, perm, inv);
    const uint32_t bytes_per_cw = (uint32_t)v2_info_bits[rate] / 8;
    uint8_t cw0[128];
    /* decodeSingleCodeword (:493-511): a fresh decoder per codeword, the first bytes_per_cw bytes */
#define V2_DECODE(idx, dst, okvar)                                                              \
    do {                                                                                        \
        float llr_[648]; uint8_t dec_[128]; int ok_ = 0, it_ = 0;                               \
        for (int j = 0; j < 648; ++j) llr_[j] = deint_bps ? soft[(size_t)(idx) * 648 + perm[j]] : soft[(size_t)(idx) * 648 + j]; \
        int nb_ = uo_ldpc_decode_soft(rate, max_iters, llr_, 648, dec_, sizeof(dec_), &ok_, &it_); \
        okvar = ok_ && nb_ >= (int)bytes_per_cw;                                                \
        if (okvar) memcpy(dst, dec_, bytes_per_cw);                                             \
    } while (0)
    int ok0;
    V2_DECODE(0, cw0, ok0);
    if (!ok0) { res[4] = 1; return 0; }
    res[3] = 1;
    int32_t h[4];
    /* RxPipeline::parseHeader (:513-527): identifyCodeword == HEADER (magic) and v2::parseHeader valid */
    if (!uo_v2_parse_header(cw0, bytes_per_cw, h) || h[1] == 0) { res[7] = 1; return 0; }
    res[2] = h[0];
    const int expected = h[1];
    if ((int)num_cw < expected) { res[5] = expected; res[7] = 2; return 0; }
    const uint32_t expected_size = h[3] ? 20u : 17u + (uint32_t)h[2] + 2u;
    /* CW0 decoded; CW1.. with the same rate; reassembleCodewords (:952-982) */
    uint32_t len = 0;
    int all_ok = 1;
    {
        uint32_t remaining = expected_size - len;
        uint32_t to_copy = remaining < bytes_per_cw ? remaining : bytes_per_cw;
        if (len + to_copy > cap) return -1;
        memcpy(frame_data + len, cw0, to_copy); len += to_copy;
    }
    for (int i = 1; i < expected; ++i) {
        uint8_t cw[128]; int ok;
        V2_DECODE(i, cw, ok);
        if (!ok) { res[4]++; all_ok = 0; continue; }
        res[3]++;
        uint32_t remaining = expected_size - len;
        if (remaining == 0) continue;
        if (bytes_per_cw >= 2 && cw[0] == 0xD5) {
            uint32_t payload = bytes_per_cw - 2, to_copy = remaining < payload ? remaining : payload;
            if (len + to_copy > cap) return -1;
            memcpy(frame_data + len, cw + 2, to_copy); len += to_copy;
        } else {
            uint32_t to_copy = remaining < bytes_per_cw ? remaining : bytes_per_cw;
            if (len + to_copy > cap) return -1;
            memcpy(frame_data + len, cw, to_copy); len += to_copy;
        }
    }
#undef V2_DECODE
    if (all_ok) { res[0] = 1; res[6] = (int32_t)len; res[7] = 4; }
    else res[7] = 3;
    return 0;
}

/* Transmit side.  frame = DataFrame::serialize (type >= 0x30: header 17 + payload + CRC 2, total_cw from
 * DataFrame::calculateCodewords(payload, rate) unless total_cw_override >= 0) or ControlFrame::serialize
 * (control types: 20 bytes, payload = 6 bytes), then encodeFrameWithLDPC(frame, rate): CW0 = first
 * bytes_per_cw bytes, CW1.. = 0xD5, index, payload; each zero padded and LDPC encoded to 81 bytes.
 * Returns the number of codewords written to codewords[n][81]. */
int uo_v2_build_frame(uint32_t rate, uint8_t type, uint8_t flags, uint16_t seq, uint32_t src_hash, uint32_t dst_hash,
                      const uint8_t* payload, uint32_t payload_len, int total_cw_override, uint8_t* codewords,
                      uint32_t cap_cw) {
    if (rate > 5 || payload_len > 4096) return -1;
    const uint32_t bytes_per_cw = (uint32_t)v2_info_bits[rate] / 8;
    uint8_t frame[17 + 4096 + 2];
    uint32_t total;
    frame[0] = 0x55; frame[1] = 0x4C; frame[2] = type; frame[3] = flags;
    frame[4] = (uint8_t)(seq >> 8); frame[5] = (uint8_t)seq;
    frame[6] = (uint8_t)(src_hash >> 16); frame[7] = (uint8_t)(src_hash >> 8); frame[8] = (uint8_t)src_hash;
    frame[9] = (uint8_t)(dst_hash >> 16); frame[10] = (uint8_t)(dst_hash >> 8); frame[11] = (uint8_t)dst_hash;
    if (v2_is_control(type)) {
        memset(frame + 12, 0, 6);
        memcpy(frame + 12, payload, payload_len < 6 ? payload_len : 6);
        uint16_t crc = uo_crc16(frame, 18);
        frame[18] = (uint8_t)(crc >> 8); frame[19] = (uint8_t)crc;
        total = 20;
    } else {
        total = 17 + payload_len + 2;
        uint32_t ncw = 1;
        if (total > bytes_per_cw) ncw = 1 + (total - bytes_per_cw + (bytes_per_cw - 2) - 1) / (bytes_per_cw - 2);
        frame[12] = (uint8_t)(total_cw_override >= 0 ? total_cw_override : (int)ncw);
        frame[13] = (uint8_t)(payload_len >> 8); frame[14] = (uint8_t)payload_len;
        uint16_t h = uo_crc16(frame, 15);
        frame[15] = (uint8_t)(h >> 8); frame[16] = (uint8_t)h;
        if (payload_len) memcpy(frame + 17, payload, payload_len);
        uint16_t f = uo_crc16(frame, total - 2);
        frame[total - 2] = (uint8_t)(f >> 8); frame[total - 1] = (uint8_t)f;
    }
    uint32_t n = 0, offset = 0;
    uint8_t chunk[128];
    while (n == 0 || offset < total) {
        if (n >= cap_cw) return -1;
        memset(chunk, 0, sizeof(chunk));
        if (n == 0) {
            memcpy(chunk, frame, total < bytes_per_cw ? total : bytes_per_cw);
            offset = bytes_per_cw;
        } else {
            chunk[0] = 0xD5; chunk[1] = (uint8_t)n;
            uint32_t remaining = total - offset, room = bytes_per_cw - 2;
            memcpy(chunk + 2, frame + offset, remaining < room ? remaining : room);
            offset += room;
        }
        if (uo_ldpc_encode(rate, chunk, bytes_per_cw, codewords + (size_t)n * 81, 81) != 81) return -1;
        ++n;
    }
    return (int)n;
}

/* ====================================================================== */
/* Modulator (stimulus), src/ofdm/modulator.cpp                            */
/* ====================================================================== */
typedef struct modulator {
    ultra_hip_config cfg; fft_plan fft; nco mixer; carriers cr; uint32_t cp;
    cf dprev[MAX_CARRIERS];
} modulator;

static int mod_init(modulator* m, const ultra_hip_config* c) {
    memset(m, 0, sizeof(*m));
    m->cfg = *c;
    if (carriers_init(&m->cr, c) != 0) return -1;
    fft_init(&m->fft, c->fft_size);
    nco_init(&m->mixer, (float)((float)c->center_freq + 0.0f), (float)c->sample_rate); /* tx_cfo_hz = 0 */
    m->cp = cyclic_prefix(c);
    return 0;
}

/* mapBits + constellation tables, src/ofdm/modulator.cpp:13-108 */
static cf map_bits(uint32_t bits, uint32_t mod) {
    const float QS = 0.7071067811865476f;
    switch (mod) {
        case ULTRA_MOD_BPSK: return (bits & 1) ? c_make(1, 0) : c_make(-1, 0);
        case ULTRA_MOD_QAM16: {
            static const float levels[] = {-3, -1, 3, 1};
            const float S = 0.3162277660168379f;
            return c_make(levels[(bits >> 2) & 3] * S, levels[bits & 3] * S);
        }
        case ULTRA_MOD_QAM32: {
            const float S = 0.1961161351381840f;
            static const float I_LEVELS[4] = {-3, -1, 1, 3};
            static const int I_GRAY[4] = {0, 1, 3, 2};
            static const float Q_LEVELS[8] = {-7, -5, -3, -1, 1, 3, 5, 7};
            static const int Q_GRAY[8] = {0, 1, 3, 2, 6, 7, 5, 4};
            int q_bits = (bits >> 2) & 7, i_bits = bits & 3, qi = 0, ii = 0;
            for (int i = 0; i < 4; ++i) if (I_GRAY[i] == i_bits) { ii = i; break; }
            for (int i = 0; i < 8; ++i) if (Q_GRAY[i] == q_bits) { qi = i; break; }
            return c_make(I_LEVELS[ii] * S, Q_LEVELS[qi] * S);
        }
        case ULTRA_MOD_QAM64: {
            static const float levels[] = {-7, -5, -1, -3, 7, 5, 1, 3};
            const float S = 0.1543033499620919f;
            return c_make(levels[(bits >> 3) & 7] * S, levels[bits & 7] * S);
        }
        case ULTRA_MOD_QAM256: {
            static const float levels[] = {-15, -13, -9, -11, -1, -3, -7, -5, 15, 13, 9, 11, 1, 3, 7, 5};
            const float S = 0.0645497224367903f;
            return c_make(levels[(bits >> 4) & 15] * S, levels[bits & 15] * S);
        }
        case ULTRA_MOD_QPSK:
        default: {
            uint32_t b = bits & 3;
            return c_make((b & 2) ? QS : -QS, (b & 1) ? QS : -QS);
        }
    }
}

/* createOFDMSymbol / createSchmidlCoxSTS + complexToReal, modulator.cpp:202-283 */
static size_t mod_emit(modulator* m, const cf* freq_domain, float* out) {
    uint32_t N = m->cfg.fft_size, cp = m->cp;
    cf td[MAX_FFT];
    memcpy(td, freq_domain, sizeof(cf) * N);
    fft_exec(&m->fft, td, 1);
    float scale = 40.0f; /* ModemConfig::output_scale default */
    size_t o = 0;
    for (uint32_t i = N - cp; i < N; ++i) { cf mixed = c_mul(td[i], nco_next(&m->mixer)); out[o++] = mixed.re * scale; }
    for (uint32_t i = 0; i < N; ++i) { cf mixed = c_mul(td[i], nco_next(&m->mixer)); out[o++] = mixed.re * scale; }
    return o;
}
static void mod_symbol_freq(const modulator* m, const cf* data_syms, int n_syms, int pilots, cf* fd) {
    for (uint32_t i = 0; i < m->cfg.fft_size; ++i) fd[i] = c_make(0, 0);
    for (int i = 0; i < m->cr.n_data && i < n_syms; ++i) fd[m->cr.data_idx[i]] = data_syms[i];
    if (pilots) for (int i = 0; i < m->cr.n_pilot; ++i) fd[m->cr.pilot_idx[i]] = m->cr.pilot_seq[i];
}

/* OFDMModulator::generatePreamble, modulator.cpp:479-532 */
static size_t mod_preamble(modulator* m, float* out) {
    uint32_t N = m->cfg.fft_size;
    m->mixer.phase = 0;
    for (int i = 0; i < m->cr.n_data; ++i) m->dprev[i] = c_make(1, 0);
    size_t guard = N + m->cp, o = 0;
    for (size_t i = 0; i < guard; ++i) out[o++] = 0.0f;
    cf fd[MAX_FFT];
    for (uint32_t i = 0; i < N; ++i) fd[i] = c_make(0, 0);
    size_t seq_idx = 0;
    for (int i = 0; i < m->cr.n_data; ++i) {
        int ci = m->cr.data_idx[i];
        if (ci % 2 == 0) fd[ci] = m->cr.sync_seq[seq_idx % (size_t)m->cr.n_sync];
        seq_idx++;
    }
    size_t n1 = mod_emit(m, fd, out + o);            /* STS once through the mixer ... */
    for (int r = 1; r < 4; ++r) memcpy(out + o + (size_t)r * n1, out + o, sizeof(float) * n1); /* ... repeated 4x */
    o += 4 * n1;
    cf lts[MAX_CARRIERS];
    for (int i = 0; i < m->cr.n_data; ++i) lts[i] = m->cr.sync_seq[(size_t)i % (size_t)m->cr.n_sync];
    mod_symbol_freq(m, lts, m->cr.n_data, 1, fd);
    size_t n2 = mod_emit(m, fd, out + o);
    memcpy(out + o + n2, out + o, sizeof(float) * n2);
    o += 2 * n2;
    return o;
}

/* OFDMModulator::generateTrainingSymbols, modulator.cpp:534-580 */
static size_t mod_training(modulator* m, int count, float* out) {
    m->mixer.phase = 0;
    for (int i = 0; i < m->cr.n_data; ++i) m->dprev[i] = c_make(1, 0);
    cf lts[MAX_CARRIERS], fd[MAX_FFT];
    for (int i = 0; i < m->cr.n_data; ++i) lts[i] = m->cr.sync_seq[(size_t)i % (size_t)m->cr.n_sync];
    size_t o = 0;
    for (int s = 0; s < count; ++s) {
        mod_symbol_freq(m, lts, m->cr.n_data, 1, fd);
        o += mod_emit(m, fd, out + o);
        for (uint32_t g = 0; g < m->cfg.symbol_guard; ++g) { out[o++] = 0; (void)nco_next(&m->mixer); }
    }
    return o;
}

/* OFDMModulator::modulate, modulator.cpp:348-477 */
static size_t mod_modulate(modulator* m, const uint8_t* data, size_t n, float* out, size_t cap) {
    uint32_t mod = m->cfg.modulation;
    size_t bpc = bits_per_symbol(mod);
    size_t cps = (size_t)m->cr.n_data;
    size_t data_idx = 0, bit_idx = 0, o = 0;
    size_t sym_len = m->cfg.fft_size + m->cp + m->cfg.symbol_guard;
    cf fd[MAX_FFT], syms[MAX_CARRIERS];
    while (data_idx < n) {
        size_t ns = 0;
        for (size_t cI = 0; cI < cps && data_idx < n; ++cI) {
            uint32_t bits = 0;
            for (size_t b = 0; b < bpc; ++b) {
                bits <<= 1;
                if (data_idx < n) {
                    bits |= (uint32_t)((data[data_idx] >> (7 - bit_idx)) & 1);
                    if (++bit_idx >= 8) { bit_idx = 0; ++data_idx; }
                }
            }
            cf s;
            if (mod == ULTRA_MOD_DBPSK) {
                cf pc = (bits & 1) ? c_make(-1, 0) : c_make(1, 0);
                s = c_mul(m->dprev[cI], pc); m->dprev[cI] = s;
            } else if (mod == ULTRA_MOD_DQPSK) {
                static const cf ph[4] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};
                s = c_mul(m->dprev[cI], ph[bits & 3]); m->dprev[cI] = s;
            } else if (mod == ULTRA_MOD_D8PSK) {
                const float pi = 3.14159265358979f;
                float angle = (float)(bits & 7) * (pi / 4.0f) + pi / 8.0f;
                s = c_mul(m->dprev[cI], c_make(cosf(angle), sinf(angle))); m->dprev[cI] = s;
            } else {
                s = map_bits(bits, mod);
            }
            syms[ns++] = s;
        }
        while (ns < cps) syms[ns++] = c_make(0, 0);
        if (o + sym_len > cap) return 0;
        mod_symbol_freq(m, syms, (int)ns, 1, fd);
        o += mod_emit(m, fd, out + o);
        for (uint32_t g = 0; g < m->cfg.symbol_guard; ++g) { out[o++] = 0; (void)nco_next(&m->mixer); }
    }
    return o;
}

int uo_modulate_frame(const ultra_hip_config* c, const uint8_t* encoded, uint32_t n_enc,
                      float* out, uint32_t cap, uint32_t* preamble_len) {
    modulator* m = (modulator*)malloc(sizeof(modulator));
    if (mod_init(m, c) != 0) { free(m); return -1; }
    size_t pre = 7 * (size_t)(c->fft_size + m->cp);
    if (pre > cap) { fft_free(&m->fft); free(m); return -1; }
    size_t p = mod_preamble(m, out);
    size_t dn = mod_modulate(m, encoded, n_enc, out + p, cap - p);
    if (preamble_len) *preamble_len = (uint32_t)p;
    fft_free(&m->fft); free(m);
    return dn ? (int)(p + dn) : -1;
}

int uo_modulate_presynced(const ultra_hip_config* c, const uint8_t* encoded, uint32_t n_enc,
                          float* out, uint32_t cap) {
    modulator* m = (modulator*)malloc(sizeof(modulator));
    if (mod_init(m, c) != 0) { free(m); return -1; }
    size_t tlen = (size_t)c->training_symbols * (c->fft_size + m->cp + c->symbol_guard);
    if (tlen > cap) { fft_free(&m->fft); free(m); return -1; }
    size_t p = mod_training(m, (int)c->training_symbols, out);
    size_t dn = mod_modulate(m, encoded, n_enc, out + p, cap - p);
    fft_free(&m->fft); free(m);
    return dn ? (int)(p + dn) : -1;
}

/* ====================================================================== */
/* Channel (stimulus)                                                      */
/* ====================================================================== */
/* counter-based generator: splitmix64 → Box-Muller.  Not the reference's
 * mt19937 + libstdc++ std::normal_distribution stream: statistically
 * equivalent stimulus only. */
typedef struct crng { uint64_t s; int have; float spare; } crng;
static inline uint64_t splitmix64(uint64_t* s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline float crng_gauss(crng* r) {
    if (r->have) { r->have = 0; return r->spare; }
    uint64_t a = splitmix64(&r->s), b = splitmix64(&r->s);
    double u1 = ((double)((a >> 11) + 1)) * (1.0 / 9007199254740993.0);
    double u2 = ((double)(b >> 11)) * (1.0 / 9007199254740992.0);
    double rad = sqrt(-2.0 * log(u1)), ang = 2.0 * M_PI * u2;
    r->spare = (float)(rad * sin(ang)); r->have = 1;
    return (float)(rad * cos(ang));
}

/* WattersonChannel::applyCFO of a freshly constructed channel (src/sim/hf_channel.hpp:95-101,161-232): the radio's
 * tuning error as the harnesses model it — mix the passband signal down from 1500 Hz, 48-tap running-mean lowpass,
 * rotate at baseband by the CFO, mix back up.  Called by process() after the noise when abs(cfo_hz) > 0.001 (:163-165);
 * buffers shorter than 256 samples pass unchanged (:173).  In place. */
int uo_channel_apply_cfo(float cfo_hz, uint32_t sample_rate, float* samples, uint32_t n) {
    if (n < 256) return 0;
    const float fc = 1500.0f;
    const float fs = (float)sample_rate;
    const float phase_inc = (float)((((double)2.0f * M_PI) * (double)cfo_hz) / (double)sample_rate);     /* :101 */
    float* I_bb = (float*)malloc(4 * (size_t)n * sizeof(float));
    if (!I_bb) return -1;
    float *Q_bb = I_bb + n, *I_filt = Q_bb + n, *Q_filt = I_filt + n;
    for (uint32_t i = 0; i < n; ++i) {
        const float t = (float)i / fs;
        const float mix_phase = (float)(((((double)2.0f * M_PI) * (double)fc)) * (double)t);
        I_bb[i] = samples[i] * cosf(mix_phase);
        Q_bb[i] = samples[i] * sinf(mix_phase);
    }
    const uint32_t win = 48;
    float I_sum = 0, Q_sum = 0;
    for (uint32_t i = 0; i < n; ++i) {
        I_sum += I_bb[i];
        Q_sum += Q_bb[i];
        if (i >= win) { I_sum -= I_bb[i - win]; Q_sum -= Q_bb[i - win]; }
        const uint32_t cnt = (i + 1 < win) ? i + 1 : win;
        I_filt[i] = I_sum / (float)cnt;
        Q_filt[i] = Q_sum / (float)cnt;
    }
    float phase = 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        const float t = (float)i / fs;
        const float mix_phase = (float)(((((double)2.0f * M_PI) * (double)fc)) * (double)t);
        const float cfo_cos = cosf(phase), cfo_sin = sinf(phase);
        const float I_cfo = I_filt[i] * cfo_cos - Q_filt[i] * cfo_sin;
        const float Q_cfo = I_filt[i] * cfo_sin + Q_filt[i] * cfo_cos;
        samples[i] = 2.0f * (I_cfo * cosf(mix_phase) - Q_cfo * sinf(mix_phase));
        phase += phase_inc;
        if ((double)phase > (double)2.0f * M_PI) phase = (float)((double)phase - (double)2.0f * M_PI);
    }
    free(I_bb);
    return 0;
}

/* WattersonChannel ctor + process, src/sim/hf_channel.hpp:66-168,258-275 */
int uo_watterson(float snr_db, float delay_ms, float doppler_hz, float g1, float g2,
                 int fading, int multipath, int noise, uint64_t seed,
                 const float* in, uint32_t n, float* out) {
    crng rng = {seed * 0x9E3779B97F4A7C15ull + 0x5EEDull, 0, 0.0f};
    uint32_t sample_rate = 48000;
    size_t delay_samples = (size_t)(delay_ms * (float)sample_rate / 1000.0f);
    float* delay_line = (float*)calloc(delay_samples + 1, sizeof(float));
    size_t dl_head = 0, dl_len = delay_samples + 1;
    float normalized_doppler = doppler_hz / (float)sample_rate;
    float fading_alpha = (float)(1.0 - exp((double)-2.0f * M_PI * (double)normalized_doppler));
    cf f1 = c_make(1.0f, 0.0f), f2 = c_make(1.0f, 0.0f);
    float input_power = 0.0f;
    for (uint32_t i = 0; i < n; ++i) input_power += in[i] * in[i];
    float input_rms = sqrtf(input_power / (float)n);
    float eff_noise = input_rms * powf(10.0f, -snr_db / 20.0f);
    for (uint32_t i = 0; i < n; ++i) {
        float sample = in[i];
        if (fading) {
            float ns = sqrtf(1.0f / fading_alpha);
            cf n1, n2;
            n1.re = ns * crng_gauss(&rng); n1.im = ns * crng_gauss(&rng);
            n2.re = ns * crng_gauss(&rng); n2.im = ns * crng_gauss(&rng);
            f1 = c_add(c_scale(f1, 1.0f - fading_alpha), c_scale(n1, fading_alpha));
            f2 = c_add(c_scale(f2, 1.0f - fading_alpha), c_scale(n2, fading_alpha));
        }
        float o = 0.0f;
        if (multipath && delay_samples > 0) {
            float h1 = fading ? c_abs(f1) : 1.0f, h2 = fading ? c_abs(f2) : 1.0f;
            o += sample * g1 * h1;
            float delayed = delay_line[dl_head];      /* deque front/pop/push of length delay+1 */
            delay_line[dl_head] = sample;
            dl_head = (dl_head + 1) % dl_len;
            o += delayed * g2 * h2;
        } else {
            float h = fading ? c_abs(f1) : 1.0f;
            o = sample * h;
        }
        if (noise) o += eff_noise * crng_gauss(&rng);
        out[i] = o;
    }
    free(delay_line);
    return 0;
}

/* ---------------------------------------------------------------------- */
/* synthetic batch (harness shape of tools/test_nvis_mode.cpp:35-93)       */
/* ---------------------------------------------------------------------- */
typedef struct mk_job {
    const ultra_hip_config* c; uint64_t seed, f0; uint32_t n0, n1; int kind;
    float snr_db, delay_ms, doppler_hz; float* audio_out; uint8_t* payload_out; uint32_t payload_bytes; int rc;
} mk_job;

static void* mk_worker(void* arg) {
    mk_job* j = (mk_job*)arg;
    const ultra_hip_config* c = j->c;
    ultra_hip_geometry g;
    if (uo_geometry(c, &g) != 0) { j->rc = -1; return NULL; }
    /* enough codewords to fill n_data_symbols; payload_out keeps the FIRST codeword's payload */
    uint32_t ncw = (g.llrs_per_frame + LDPC_N - 1) / LDPC_N;
    uint32_t tx_symbols = (ncw * LDPC_N + 7 + g.llrs_per_symbol - 1) / g.llrs_per_symbol + 1;
    uint32_t cap = (8 + c->training_symbols + tx_symbols) * (g.symbol_samples + 64);
    float* sig = (float*)malloc(sizeof(float) * cap);
    float* chan = (float*)malloc(sizeof(float) * cap);
    uint32_t nraw = ncw * j->payload_bytes;
    uint8_t* raw = (uint8_t*)malloc(nraw);
    uint8_t* enc = (uint8_t*)malloc((size_t)(ncw + 1) * 96);
    for (uint32_t q = j->n0; q < j->n1; ++q) {
        uint64_t f = j->f0 + q;
        uint8_t* pl = j->payload_out + (size_t)q * j->payload_bytes;
        uint64_t s = (j->seed ^ f) * 0xD1342543DE82EF95ull + 0x5EEDull;
        for (uint32_t b = 0; b < nraw; ++b) raw[b] = (uint8_t)(splitmix64(&s) >> 56);
        memcpy(pl, raw, j->payload_bytes);
        int ne = uo_ldpc_encode(c->code_rate, raw, nraw, enc, (ncw + 1) * 96);
        if (ne <= 0) { j->rc = -1; break; }
        uint32_t pre = 0; int total;
        if (c->entry == ULTRA_ENTRY_PRESYNCED) { total = uo_modulate_presynced(c, enc, (uint32_t)ne, sig, cap); pre = 0; }
        else total = uo_modulate_frame(c, enc, (uint32_t)ne, sig, cap, &pre);
        if (total <= 0 || (uint32_t)total < pre + g.frame_samples) { j->rc = -2; break; }
        float mx = 0;
        for (int i = 0; i < total; ++i) { float a = fabsf(sig[i]); mx = f_max(mx, a); }
        for (int i = 0; i < total; ++i) sig[i] *= 0.5f / mx;
        const float* src = sig;
        if (j->kind == 1) {          /* AWGN drawn as tools/test_nvis_mode.cpp:78-86 */
            float sp = 0; for (int i = 0; i < total; ++i) sp += sig[i] * sig[i];
            sp /= (float)total;
            float nstd = sqrtf(sp / powf(10.0f, j->snr_db / 10.0f));
            crng r = {(j->seed ^ f) * 0x9E3779B97F4A7C15ull + 0xA5A5ull, 0, 0.0f};
            for (int i = 0; i < total; ++i) chan[i] = sig[i] + nstd * crng_gauss(&r);
            src = chan;
        } else if (j->kind == 2) {   /* Watterson, fading restarted per frame */
            uo_watterson(j->snr_db, j->delay_ms, j->doppler_hz, 0.707f, 0.707f, 1, 1, 1, j->seed ^ (f * 0x100000001B3ull),
                         sig, (uint32_t)total, chan);
            src = chan;
        }
        memcpy(j->audio_out + (size_t)q * g.frame_samples, src + pre, sizeof(float) * g.frame_samples);
    }
    free(sig); free(chan); free(raw); free(enc);
    return NULL;
}

int uo_make_batch(const ultra_hip_config* c, uint64_t seed, uint64_t f0, uint32_t n, int n_threads,
                  int channel_kind, float snr_db, float delay_ms, float doppler_hz,
                  float* audio_out, uint8_t* payload_out, uint32_t payload_bytes) {
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > n) n_threads = n ? (int)n : 1;
    (void)ldpc_get(c->code_rate);
    mk_job* jobs = (mk_job*)calloc((size_t)n_threads, sizeof(mk_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; ++t) {
        mk_job* j = &jobs[t];
        j->c = c; j->seed = seed; j->f0 = f0; j->kind = channel_kind; j->snr_db = snr_db;
        j->delay_ms = delay_ms; j->doppler_hz = doppler_hz; j->audio_out = audio_out;
        j->payload_out = payload_out; j->payload_bytes = payload_bytes;
        j->n0 = (uint32_t)((uint64_t)n * (uint64_t)t / (uint64_t)n_threads);
        j->n1 = (uint32_t)((uint64_t)n * (uint64_t)(t + 1) / (uint64_t)n_threads);
        if (n_threads == 1) mk_worker(j); else pthread_create(&th[t], NULL, mk_worker, j);
    }
    int rc = 0;
    for (int t = 0; t < n_threads; ++t) { if (n_threads > 1) pthread_join(th[t], NULL); if (jobs[t].rc) rc = jobs[t].rc; }
    free(jobs); free(th);
    return rc;
}

/* ---------------------------------------------------------------------- */
/* BPSK-over-AWGN LLR stimulus — twin of the device's   */
/* ultra_hip_make_llr_batch.  Build-defined (the reference has no LDPC-only */
/* SNR harness); harness shape of tools/test_mode_snr.cpp:18-109: random    */
/* payload of floor(k/8) bytes -> LDPCEncoder::encode (ldpc_encoder.cpp:    */
/* 193-257) -> noise -> LLR.  Every step is ONE IEEE float operation or a   */
/* libm call (logf, sqrtf, sinf, cosf), so the device, whose kernels carry  */
/* bit-exact restatements of those libm functions, produces the same bits.  */
/* ---------------------------------------------------------------------- */
static inline uint64_t splitmix_at(uint64_t s0, uint64_t n) {      /* output of call n (0-based) of splitmix64 seeded s0 */
    uint64_t z = s0 + (n + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline void gauss_pair_exact(uint64_t key, uint64_t n, float* g0, float* g1) {
    uint64_t z = splitmix_at(key, n);
    float u1 = ((float)((z >> 40) & 0xFFFFFFull) + 1.0f) * (1.0f / 16777216.0f);   /* (0, 1] */
    float u2 = (float)((z >> 8) & 0xFFFFFFull) * (1.0f / 16777216.0f);             /* [0, 1) */
    float rad = sqrtf(-2.0f * logf(u1));
    float ang = 6.283185307179586f * u2;
    *g0 = rad * cosf(ang);
    *g1 = rad * sinf(ang);
}
int uo_make_llr_batch(uint32_t rate, uint64_t seed, uint64_t c0, uint32_t n_cw, float esn0_db,
                      float* llr_out, uint8_t* payload_out) {
    const ldpc_code* code = ldpc_get(rate);
    int k = code->k, pb = k / 8;
    double esn0 = pow(10.0, (double)esn0_db / 10.0);
    float sigma2 = (float)(1.0 / (2.0 * esn0));
    float sigma = (float)sqrt(1.0 / (2.0 * esn0));
    uint8_t enc[96];
    for (uint32_t w = 0; w < n_cw; ++w) {
        uint64_t c = c0 + w;
        uint64_t s0 = (seed ^ c) * 0xD1342543DE82EF95ull + 0x5EEDull;
        uint8_t* pl = payload_out + (size_t)w * (size_t)pb;
        for (int b = 0; b < pb; ++b) pl[b] = (uint8_t)(splitmix_at(s0, (uint64_t)b) >> 56);
        if (uo_ldpc_encode(rate, pl, (uint32_t)pb, enc, sizeof(enc)) != 81) return -1;
        uint64_t key = ((seed ^ (c * 0x100000001B3ull)) * 0x9E3779B97F4A7C15ull + 0xC4A77E1ull) ^ 0x4444ull;
        float* out = llr_out + (size_t)LDPC_N * w;
        for (int p = 0; p < LDPC_N / 2; ++p) {
            float g0, g1;
            gauss_pair_exact(key, (uint64_t)p, &g0, &g1);
            int b0 = (enc[(2 * p) / 8] >> (7 - (2 * p) % 8)) & 1, b1 = (enc[(2 * p + 1) / 8] >> (7 - (2 * p + 1) % 8)) & 1;
            float x0 = b0 ? -1.0f : 1.0f, x1 = b1 ? -1.0f : 1.0f;
            float y0 = x0 + sigma * g0, y1 = x1 + sigma * g1;
            out[2 * p] = (2.0f * y0) / sigma2;
            out[2 * p + 1] = (2.0f * y1) / sigma2;
        }
    }
    return 0;
}
