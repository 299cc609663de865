// witness.hip -- trace (witness) generation kernels for the tables whose rows are data-parallel.
//
// Every kernel writes the table column-major (the flattened Vec<PolynomialValues<F>> of util.rs:37-46) with consecutive
// threads on consecutive rows, so each column store is a contiguous 512 B run per wavefront.  References are given at
// each kernel.  The Memory table's rows come sorted from the CPU-side generator (as in the reference) and have no kernel.
#include "poseidon_dev.h"
#include "hash_constants_dev.h"
#include "zkm_internal.h"

// ------------------------------------------------------------------ KeccakSpongeStark witness (a12)
// Rows of one operation chain through the permutation (keccak_sponge_stark.rs:253-299: the state after row k is the input of row
// k + 1), rows of different operations are independent; column map keccak_sponge/columns.rs:19-70.  Two launches:
//   (1) k_keccak_sponge_states: one lane per OPERATION walks its chain and parks the sponge state BEFORE each of its rows (25 words)
//       plus the row's operation index -- 204 B per row, the only data that depends on the chain;
//   (2) k_keccak_sponge_rows: one lane per ROW rebuilds the row from that state (one more permutation: 1 M extra Keccak-f cost
//       0.25 ms) and stores all 470 cells, zeros included -- lane = row, so every column store is one contiguous 512 B wave access
//       and the table needs no zero-fill.  (The one-lane-per-operation form stored only the non-zero cells of a zero-filled table,
//       4.6 rows apart per lane: 8-byte partial-line writes, 14 GB of HBM traffic for a 3.9 GB table -- profiles/r03_a_*.)
// word i of the padded 136-byte block of a row (pad10*1 on the final row, :334-341)
__device__ __forceinline__ uint64_t sponge_block_word(const uint8_t* __restrict__ msg, size_t rem, bool full, int i) {
    uint64_t w = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const size_t pos = (size_t)8 * i + j;
        uint32_t b = pos < rem ? msg[pos] : 0;
        if (!full) {
            if (pos == rem) b = (rem == 135) ? 0x81 : 0x01;
            else if (pos == 135) b = 0x80;
        }
        w |= (uint64_t)b << (8 * j);
    }
    return w;
}

__global__ __launch_bounds__(128) void k_keccak_sponge_states(const uint8_t* __restrict__ inputs, const uint64_t* __restrict__ off,
                                                              const uint64_t* __restrict__ row_off, size_t nops,
                                                              uint64_t* __restrict__ row_state, uint32_t* __restrict__ row_op) {
    const size_t op = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (op >= nops) return;
    const uint8_t* msg = inputs + off[op];
    const size_t len = off[op + 1] - off[op];
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    size_t row = row_off[op], absorbed = 0;
    for (;;) {
        const size_t rem = len - absorbed;
        const bool full = rem >= 136;
        uint64_t* rs = row_state + row * 25;
#pragma unroll
        for (int i = 0; i < 25; i++) rs[i] = st[i];
        row_op[row] = (uint32_t)op;
        if (!full) break;                                   // (the state after the final row is not needed by anyone)
#pragma unroll
        for (int i = 0; i < 17; i++) st[i] ^= sponge_block_word(msg + absorbed, rem, full, i);
        keccakf_dev(st);
        row++;
        absorbed += 136;
    }
}

__global__ __launch_bounds__(256) void k_keccak_sponge_rows(const uint8_t* __restrict__ inputs, const uint64_t* __restrict__ off,
                                                            const uint64_t* __restrict__ meta, const uint64_t* __restrict__ row_off,
                                                            const uint64_t* __restrict__ row_state, const uint32_t* __restrict__ row_op,
                                                            size_t rows_used, size_t n, gl_t* __restrict__ out) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    gl_t* o = out + r;
    if (r >= rows_used) {                                   // padding rows are all zero (keccak_sponge_stark.rs:243-247)
#pragma unroll 10
        for (int cidx = 0; cidx < ZKM_KECCAK_SPONGE_COLS; cidx++) o[(size_t)cidx * n] = 0;
        return;
    }
    const size_t op = row_op[r];
    const size_t absorbed = (r - row_off[op]) * 136;
    const uint8_t* msg = inputs + off[op] + absorbed;
    const size_t len = off[op + 1] - off[op], nwords = (len + 3) / 4, rem = len - absorbed;
    const bool full = rem >= 136;
    o[0] = full ? 1 : 0;
    o[1 * n] = meta[4 * op];
    o[2 * n] = meta[4 * op + 1];
    {
        const uint64_t vbase = meta[4 * op + 2];
#pragma unroll 2
        for (size_t i = 0; i < 34; i++) {
            const size_t w = absorbed / 4 + i;
            o[(3 + i) * n] = w < nwords ? vbase + w : 0;
        }
    }
    o[37 * n] = meta[4 * op + 3];
    o[38 * n] = len;
    o[39 * n] = absorbed;
#pragma unroll 8
    for (size_t k = 0; k < 136; k++) o[(40 + k) * n] = (!full && k == rem) ? 1 : 0;
    uint64_t st[25];
    {
        const uint64_t* rs = row_state + r * 25;
#pragma unroll
        for (int i = 0; i < 25; i++) st[i] = rs[i];
    }
#pragma unroll
    for (int i = 0; i < 34; i++) o[(size_t)(176 + i) * n] = (uint32_t)(st[i / 2] >> (32 * (i & 1)));
#pragma unroll
    for (int i = 0; i < 16; i++) o[(size_t)(210 + i) * n] = (uint32_t)(st[(34 + i) / 2] >> (32 * ((34 + i) & 1)));
#pragma unroll
    for (int i = 0; i < 17; i++) {
        const uint64_t w = sponge_block_word(msg, rem, full, i);
#pragma unroll
        for (int j = 0; j < 8; j++) o[(size_t)(226 + 8 * i + j) * n] = (uint8_t)(w >> (8 * j));
        st[i] ^= w;
    }
#pragma unroll
    for (int i = 0; i < 34; i++) o[(size_t)(362 + i) * n] = (uint32_t)(st[i / 2] >> (32 * (i & 1)));
    keccakf_dev(st);
#pragma unroll
    for (int i = 0; i < 42; i++) o[(size_t)(396 + i) * n] = (uint32_t)(st[(8 + i) / 2] >> (32 * ((8 + i) & 1)));
#pragma unroll
    for (int i = 0; i < 32; i++) o[(size_t)(438 + i) * n] = (uint8_t)(st[i / 8] >> (8 * (i & 7)));
}

void zkm_launch_keccak_sponge_trace(zkm_ctx* c, const uint8_t* d_inputs, const uint64_t* d_off, const uint64_t* d_meta,
                                    const uint64_t* d_row_off, size_t nops, size_t rows_used, unsigned log_n, gl_t* out) {
    const size_t n = (size_t)1 << log_n;
    zkm_scratch row_state(c, (rows_used ? rows_used : 1) * 25 * sizeof(uint64_t)), row_op(c, (rows_used ? rows_used : 1) * sizeof(uint32_t));
    zkm_prof_scope ps(c, "keccak_sponge_trace");
    if (nops) hipLaunchKernelGGL(k_keccak_sponge_states, dim3((nops + 127) / 128), dim3(128), 0, c->stream, d_inputs, d_off, d_row_off, nops,
                                 row_state.as<uint64_t>(), row_op.as<uint32_t>());
    hipLaunchKernelGGL(k_keccak_sponge_rows, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_inputs, d_off, d_meta, d_row_off,
                       row_state.as<uint64_t>(), row_op.as<uint32_t>(), rows_used, n, out);
    ZKM_HIP_CHECK(hipGetLastError());
}


// ------------------------------------------------------------------ PoseidonStark witness (a13)
__device__ __forceinline__ uint64_t splitmix_at(uint64_t seed, uint64_t k) {
    uint64_t z = seed + k * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// One row per lane; every column store is a contiguous wave access (column-major output).
// Column map: poseidon/columns.rs:3-54 (FILTER 0, in 1..12, out 13..24, TIMESTAMP 25, full0 26.., partial 122.., full1 166..)
__global__ __launch_bounds__(256) void k_poseidon_trace(uint64_t seed, const uint64_t* __restrict__ inputs, const uint64_t* __restrict__ ts,
                                                        size_t num_perms, size_t n, gl_t* __restrict__ out) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    bool real = r < num_perms;
    uint64_t s[12];
    gl_t* o = out + r;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        s[i] = !real ? 0 : inputs ? gl_canon(inputs[r * 12 + i]) : gl_canon(splitmix_at(seed, r * 12 + i + 1));
        o[(1 + i) * n] = s[i];
    }
    o[0] = real ? 1 : 0;
    o[25 * n] = (real && ts) ? ts[r] : 0;
    int rc = 0;
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        int base = half == 0 ? 26 : 166;
#pragma unroll 1
        for (int rr = 0; rr < 4; rr++, rc++) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                uint64_t x = gl_add_loose(s[i], PC::ZKM_POSEIDON_RC[rc * 12 + i]);
                gl_t x3 = gl_mul(gl_mul_loose(x, x), x);
                gl_t x7 = gl_mul(x, gl_mul_loose(x3, x3));
                o[(size_t)(base + 24 * rr + 2 * i) * n] = x3;
                o[(size_t)(base + 24 * rr + 2 * i + 1) * n] = x7;
                s[i] = x7;
            }
            poseidon_mds(s);
        }
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) s[i] = gl_add_loose(s[i], PC::ZKM_POSEIDON_FAST_FIRST_RC[i]);
            uint64_t t[12];
            t[0] = s[0];
#pragma unroll
            for (int c = 1; c < 12; c++) {
                uint64_t acc = 0;
#pragma unroll
                for (int q = 1; q < 12; q++) acc = gl_add_loose(acc, gl_mul_loose(s[q], PC::ZKM_POSEIDON_FAST_INIT[q - 1][c - 1]));
                t[c] = acc;
            }
#pragma unroll
            for (int i = 0; i < 12; i++) s[i] = t[i];
#pragma unroll 1
            for (int q = 0; q < 22; q++) {
                uint64_t x = s[0];
                gl_t x3 = gl_mul(gl_mul_loose(x, x), x);
                gl_t x7 = gl_mul(x, gl_mul_loose(x3, x3));
                o[(size_t)(122 + 2 * q) * n] = x3;
                o[(size_t)(122 + 2 * q + 1) * n] = x7;
                uint64_t s0 = gl_add_loose(x7, PC::ZKM_POSEIDON_FAST_RC[q]);
                uint64_t d = gl_mul_loose(s0, 25);
#pragma unroll
                for (int i = 1; i < 12; i++) d = gl_add_loose(d, gl_mul_loose(s[i], PC::ZKM_POSEIDON_FAST_W_HATS[q][i - 1]));
#pragma unroll
                for (int i = 1; i < 12; i++) s[i] = gl_add_loose(s[i], gl_mul_loose(s0, PC::ZKM_POSEIDON_FAST_VS[q][i - 1]));
                s[0] = d;
            }
            rc += 22;
        }
    }
#pragma unroll
    for (int i = 0; i < 12; i++) o[(size_t)(13 + i) * n] = gl_canon(s[i]);
}

void zkm_launch_poseidon_trace(zkm_ctx* c, uint64_t seed, const uint64_t* inputs, const uint64_t* ts, size_t num_perms, unsigned log_n,
                               gl_t* out) {
    size_t n = (size_t)1 << log_n;
    zkm_prof_scope ps(c, "poseidon_trace");
    hipLaunchKernelGGL(k_poseidon_trace, dim3((n + 255) / 256), dim3(256), 0, c->stream, seed, inputs, ts, num_perms, n, out);
    ZKM_HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ LogicStark witness (logic.rs:122-183)
// One thread per row; stores are coalesced per column (column-major).
__global__ __launch_bounds__(256) void k_logic_trace(const uint32_t* __restrict__ ops, size_t nops, size_t n, gl_t* __restrict__ out,
                                                     int* __restrict__ bad) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint32_t op = 4, a = 0, b = 0, res = 0;
    if (r < nops) {
        op = ops[3 * r];
        a = ops[3 * r + 1];
        b = ops[3 * r + 2];
        if (op > 3) { *bad = 1; op = 4; a = b = 0; }
        else res = op == 0 ? (a & b) : op == 1 ? (a | b) : op == 2 ? (a ^ b) : ~(a | b);
    }
#pragma unroll
    for (uint32_t f = 0; f < 4; f++) out[(size_t)f * n + r] = op == f;
#pragma unroll 8
    for (int i = 0; i < 32; i++) {
        out[(size_t)(4 + i) * n + r] = (a >> i) & 1;
        out[(size_t)(36 + i) * n + r] = (b >> i) & 1;
    }
    out[(size_t)68 * n + r] = res;
}

void zkm_launch_logic_trace(zkm_ctx* c, const uint32_t* d_ops, size_t nops, size_t n, gl_t* out, int* d_bad) {
    zkm_prof_scope ps(c, "logic_trace");
    hipLaunchKernelGGL(k_logic_trace, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_ops, nops, n, out, d_bad);
    ZKM_HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ KeccakStark witness (keccak/keccak_stark.rs:62-226)
// One thread per trace row (permutation p, round r): replays r rounds from the input (at most 23 cheap rounds; the row's
// 2431 stores dominate), then emits the round's registers.  Consecutive threads own consecutive rows, so every
// column store is a contiguous 512-byte run per wavefront.  State index: a[x + 5y] = A(x, y).
__global__ __launch_bounds__(256) void k_keccak_trace(const uint64_t* __restrict__ inputs, const uint64_t* __restrict__ ts, size_t nperms,
                                                      size_t n, gl_t* __restrict__ out) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    gl_t* o = out + row;
    size_t p = row / 24;
    int round = (int)(row - p * 24);
    if (p >= nperms) {  // padding rows are all-zero (keccak_stark.rs:77-79)
        for (int c = 0; c < ZKM_KECCAK_COLS; c++) o[(size_t)c * n] = 0;
        return;
    }
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = inputs[25 * p + i];
#pragma unroll 1
    for (int r = 0; r < round; r++) keccak_round_dev(a, KECCAK_RC_DEV[r]);
    for (int i = 0; i < 24; i++) o[(size_t)i * n] = i == round;
    o[(size_t)24 * n] = ts[p];
    uint64_t c[5], cp[5];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
    for (int x = 0; x < 5; x++) cp[x] = c[x] ^ c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
#pragma unroll
    for (int x = 0; x < 5; x++) {
#pragma unroll
        for (int y = 0; y < 5; y++) {
            o[(size_t)(25 + (x * 5 + y) * 2) * n] = (uint32_t)a[x + 5 * y];
            o[(size_t)(25 + (x * 5 + y) * 2 + 1) * n] = a[x + 5 * y] >> 32;
        }
#pragma unroll 8
        for (int z = 0; z < 64; z++) {
            o[(size_t)(75 + x * 64 + z) * n] = (c[x] >> z) & 1;
            o[(size_t)(395 + x * 64 + z) * n] = (cp[x] >> z) & 1;
        }
    }
    // A' = A ^ C ^ C' (theta), written bit by bit; then the round proper gives A'' (before iota) and the iota output
#pragma unroll
    for (int x = 0; x < 5; x++)
#pragma unroll
        for (int y = 0; y < 5; y++) {
            uint64_t ap = a[x + 5 * y] ^ c[x] ^ cp[x];
#pragma unroll 8
            for (int z = 0; z < 64; z++) o[(size_t)(715 + x * 320 + y * 64 + z) * n] = (ap >> z) & 1;
        }
    keccak_round_dev(a, 0);
#pragma unroll
    for (int x = 0; x < 5; x++)
#pragma unroll
        for (int y = 0; y < 5; y++) {
            o[(size_t)(2315 + x * 10 + y * 2) * n] = (uint32_t)a[x + 5 * y];
            o[(size_t)(2315 + x * 10 + y * 2 + 1) * n] = a[x + 5 * y] >> 32;
        }
#pragma unroll 8
    for (int z = 0; z < 64; z++) o[(size_t)(2365 + z) * n] = (a[0] >> z) & 1;
    uint64_t appp = a[0] ^ KECCAK_RC_DEV[round];
    o[(size_t)2429 * n] = (uint32_t)appp;
    o[(size_t)2430 * n] = appp >> 32;
}

void zkm_launch_keccak_trace(zkm_ctx* c, const uint64_t* d_inputs, const uint64_t* d_ts, size_t nperms, size_t n, gl_t* out) {
    zkm_prof_scope ps(c, "keccak_trace");
    hipLaunchKernelGGL(k_keccak_trace, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_inputs, d_ts, nperms, n, out);
    ZKM_HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ PoseidonSpongeStark witness (poseidon_sponge_stark.rs:186-381)
// One lane per sponge operation, as k_keccak_sponge_trace; column map poseidon_sponge/columns.rs:17-66.  The output buffer is
// zero-filled first; only non-zero cells are stored.
__global__ __launch_bounds__(128) void k_poseidon_sponge_trace(const uint8_t* __restrict__ inputs, const uint64_t* __restrict__ off,
                                                               const uint64_t* __restrict__ meta, const uint64_t* __restrict__ row_off,
                                                               size_t nops, size_t n, gl_t* __restrict__ out) {
    size_t op = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (op >= nops) return;
    const uint8_t* msg = inputs + off[op];
    const size_t len = off[op + 1] - off[op], nwords = (len + 3) / 4;
    const uint64_t ctxv = meta[4 * op], seg = meta[4 * op + 1], vbase = meta[4 * op + 2], ts = meta[4 * op + 3];
    uint64_t st[12];
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = 0;
    size_t row = row_off[op], absorbed = 0;
    for (;;) {
        const size_t rem = len - absorbed;
        const bool full = rem >= 32;
        gl_t* o = out + row;
        if (full) o[0] = 1;
        else o[(size_t)(14 + rem) * n] = 1;
        o[1 * n] = ctxv;
        o[2 * n] = seg;
        for (size_t i = 0; i < 8; i++) {
            size_t w = absorbed / 4 + i;
            if (w < nwords) o[(3 + i) * n] = vbase + w;
        }
        o[11 * n] = ts;
        o[12 * n] = len;
        o[13 * n] = absorbed;
#pragma unroll
        for (int i = 0; i < 12; i++) o[(size_t)(46 + i) * n] = st[i];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t w = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                size_t pos = (size_t)4 * i + j;
                uint32_t b = pos < rem ? msg[absorbed + pos] : 0;
                if (!full) {
                    if (pos == rem) b = (rem == 31) ? 0x81 : 0x01;
                    else if (pos == 31) b = 0x80;
                }
                if (b) o[(size_t)(58 + pos) * n] = b;
                w |= (uint64_t)b << (8 * j);
            }
            st[i] = w;
            o[(size_t)(90 + i) * n] = w;
        }
        poseidon_permute(st);
#pragma unroll
        for (int i = 0; i < 8; i++) o[(size_t)(98 + i) * n] = st[4 + i];
#pragma unroll
        for (int i = 0; i < 4; i++) o[(size_t)(106 + i) * n] = st[i];
        row++;
        if (!full) break;
        absorbed += 32;
    }
}

void zkm_launch_poseidon_sponge_trace(zkm_ctx* c, const uint8_t* d_inputs, const uint64_t* d_off, const uint64_t* d_meta,
                                      const uint64_t* d_row_off, size_t nops, unsigned log_n, gl_t* out) {
    size_t n = (size_t)1 << log_n;
    ZKM_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)ZKM_POSEIDON_SPONGE_COLS * n * sizeof(gl_t), c->stream));
    if (!nops) return;
    zkm_prof_scope ps(c, "poseidon_sponge_trace");
    hipLaunchKernelGGL(k_poseidon_sponge_trace, dim3((nops + 127) / 128), dim3(128), 0, c->stream, d_inputs, d_off, d_meta, d_row_off, nops, n, out);
    ZKM_HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ SHA-256 message-schedule witnesses
// ShaExtendStark::generate_trace (sha_extend/sha_extend_stark.rs:121-236) and ShaExtendSpongeStark::generate_trace
// (sha_extend_sponge/sha_extend_sponge_stark.rs:131-215).  One thread per row, column-major coalesced stores.
__device__ __forceinline__ uint32_t rotr32_dev(uint32_t x, unsigned r) { return (x >> r) | (x << (32 - r)); }
__device__ __forceinline__ void put_le4_dev(gl_t* o, size_t n, int col, uint32_t v) {
#pragma unroll
    for (int j = 0; j < 4; j++) o[(size_t)(col + j) * n] = (v >> (8 * j)) & 0xFF;
}
__device__ __forceinline__ void put_rot_dev(gl_t* o, size_t n, int col, uint32_t in, unsigned r, bool is_shift) {
    uint32_t shift = in >> r, carry = in & ((1u << r) - 1);
    put_le4_dev(o, n, col, is_shift ? shift : rotr32_dev(in, r));
    o[(size_t)(col + 4) * n] = shift;
    o[(size_t)(col + 5) * n] = carry;
}
__global__ __launch_bounds__(256) void k_sha_extend_trace(const uint8_t* __restrict__ inputs, const uint64_t* __restrict__ ts, size_t k,
                                                          size_t n, gl_t* __restrict__ out) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    gl_t* o = out + r;
    if (r >= k) {
        for (int c = 0; c < ZKM_SHA_EXTEND_COLS; c++) o[(size_t)c * n] = 0;
        return;
    }
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint8_t* b = inputs + 16 * r + 4 * q;
        w[q] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        put_le4_dev(o, n, 8 + 4 * q, w[q]);
    }
    const uint32_t w15 = w[0], w2 = w[1], w16 = w[2], w7 = w[3];
    put_rot_dev(o, n, 40, w15, 7, false);
    put_rot_dev(o, n, 46, w15, 18, false);
    put_rot_dev(o, n, 70, w15, 3, true);
    const uint32_t s0i = rotr32_dev(w15, 7) ^ rotr32_dev(w15, 18), s0 = s0i ^ (w15 >> 3);
    put_le4_dev(o, n, 24, s0i);
    put_le4_dev(o, n, 28, s0);
    put_rot_dev(o, n, 52, w2, 17, false);
    put_rot_dev(o, n, 58, w2, 19, false);
    put_rot_dev(o, n, 64, w2, 10, true);
    const uint32_t s1i = rotr32_dev(w2, 17) ^ rotr32_dev(w2, 19), s1 = s1i ^ (w2 >> 10);
    put_le4_dev(o, n, 32, s1i);
    put_le4_dev(o, n, 36, s1);
    const uint64_t wide = (uint64_t)s1 + w7 + s0 + w16;
    put_le4_dev(o, n, 0, (uint32_t)wide);
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) o[(size_t)(4 + c) * n] = (uint32_t)(wide >> 32) == c;
    o[(size_t)76 * n] = ts[r];
    o[(size_t)77 * n] = 1;
}

// row = 48 e + round; the thread recomputes the schedule of its block up to its round (at most 48 cheap steps)
__global__ __launch_bounds__(256) void k_sha_extend_sponge_trace(const uint32_t* __restrict__ w16, const uint64_t* __restrict__ meta, size_t k,
                                                                 size_t n, gl_t* __restrict__ out) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    gl_t* o = out + row;
    size_t e = row / 48;
    int rd = (int)(row - e * 48);
    if (e >= k) {
        for (int c = 0; c < ZKM_SHA_EXTEND_SPONGE_COLS; c++) o[(size_t)c * n] = 0;
        return;
    }
    uint32_t w[16];  // sliding window: w[j & 15] holds word j
#pragma unroll
    for (int j = 0; j < 16; j++) w[j] = w16[16 * e + j];
    uint32_t in[4] = {0, 0, 0, 0}, outw = 0;
    for (int i = 16; i <= rd + 16; i++) {
        uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15], w16v = w[(i - 16) & 15], w7 = w[(i - 7) & 15];
        uint32_t s0 = rotr32_dev(w15, 7) ^ rotr32_dev(w15, 18) ^ (w15 >> 3);
        uint32_t s1 = rotr32_dev(w2, 17) ^ rotr32_dev(w2, 19) ^ (w2 >> 10);
        outw = s1 + w16v + s0 + w7;
        in[0] = w15; in[1] = w2; in[2] = w16v; in[3] = w7;
        w[i & 15] = outw;
    }
    for (int i = 0; i < 48; i++) o[(size_t)i * n] = i == rd;
    const int i = rd + 16;
    const int src[4] = {i - 15, i - 2, i - 16, i - 7};
    const uint64_t base = meta[4 * e + 2];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        put_le4_dev(o, n, 48 + 4 * q, in[q]);
        o[(size_t)(68 + q) * n] = base + 4 * (uint64_t)src[q];
    }
    put_le4_dev(o, n, 64, outw);
    o[(size_t)72 * n] = base + 4 * (uint64_t)i;
    o[(size_t)73 * n] = meta[4 * e];
    o[(size_t)74 * n] = meta[4 * e + 1];
    o[(size_t)75 * n] = meta[4 * e + 3] + 20 * (uint64_t)rd;
}

void zkm_launch_sha_extend_trace(zkm_ctx* c, const uint8_t* d_inputs, const uint64_t* d_ts, size_t k, size_t n, gl_t* out) {
    zkm_prof_scope ps(c, "sha_extend_trace");
    hipLaunchKernelGGL(k_sha_extend_trace, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_inputs, d_ts, k, n, out);
    ZKM_HIP_CHECK(hipGetLastError());
}
void zkm_launch_sha_extend_sponge_trace(zkm_ctx* c, const uint32_t* d_w16, const uint64_t* d_meta, size_t k, size_t n, gl_t* out) {
    zkm_prof_scope ps(c, "sha_extend_sponge_trace");
    hipLaunchKernelGGL(k_sha_extend_sponge_trace, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_w16, d_meta, k, n, out);
    ZKM_HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------ SHA-256 compression witnesses
// ShaCompressStark::generate_trace (sha_compress/sha_compress_stark.rs:227-400, rows as emitted by witness/util.rs:605-690: 65 per
// compression) and ShaCompressSpongeStark::generate_trace (sha_compress_sponge_stark.rs:118-230).  One thread per row; the thread
// replays the rounds before its own (at most 64 cheap steps).
__device__ __forceinline__ void sha_round_dev(uint32_t (&s)[8], uint32_t w, uint32_t kc) {
    uint32_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
    uint32_t t1 = h + (rotr32_dev(e, 6) ^ rotr32_dev(e, 11) ^ rotr32_dev(e, 25)) + ((e & f) ^ (~e & g)) + kc + w;
    uint32_t t2 = (rotr32_dev(a, 2) ^ rotr32_dev(a, 13) ^ rotr32_dev(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    s[7] = g; s[6] = f; s[5] = e; s[4] = d + t1; s[3] = c; s[2] = b; s[1] = a; s[0] = t1 + t2;
}
template <int NC>
__device__ __forceinline__ void put_wadd_dev(gl_t* o, size_t n, int col, uint64_t wide) {
    put_le4_dev(o, n, col, (uint32_t)wide);
#pragma unroll
    for (uint32_t c = 0; c < NC; c++) o[(size_t)(col + 4 + c) * n] = (uint32_t)(wide >> 32) == c;
}
__global__ __launch_bounds__(256) void k_sha_compress_trace(const uint32_t* __restrict__ hx, const uint32_t* __restrict__ w,
                                                            const uint64_t* __restrict__ meta, size_t k, size_t n, gl_t* __restrict__ out) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    gl_t* o = out + row;
    size_t e_ = row / 65;
    int rd = (int)(row - e_ * 65);
    if (e_ >= k) {
        for (int c = 0; c < ZKM_SHA_COMPRESS_COLS; c++) o[(size_t)c * n] = 0;
        return;
    }
    uint32_t s[8];
#pragma unroll
    for (int q = 0; q < 8; q++) s[q] = hx[8 * e_ + q];
    for (int i = 0; i < rd; i++) sha_round_dev(s, w[64 * e_ + i], SHA256_K_DEV[i]);
    const uint32_t wi = rd < 64 ? w[64 * e_ + rd] : 0, ki = rd < 64 ? SHA256_K_DEV[rd] : 0;
    const uint32_t a = s[0], b = s[1], c = s[2], d = s[3], e = s[4], f = s[5], g = s[6], h = s[7];
#pragma unroll
    for (int q = 0; q < 8; q++) put_le4_dev(o, n, 4 * q, s[q]);
    put_le4_dev(o, n, 32, ~e);
    put_le4_dev(o, n, 36, wi);
    put_le4_dev(o, n, 40, ki);
    const uint32_t s1i = rotr32_dev(e, 6) ^ rotr32_dev(e, 11), s1 = s1i ^ rotr32_dev(e, 25), eaf = e & f, eng = ~e & g, ch = eaf ^ eng;
    put_le4_dev(o, n, 44, s1i); put_le4_dev(o, n, 48, s1); put_le4_dev(o, n, 52, eaf); put_le4_dev(o, n, 56, eng); put_le4_dev(o, n, 60, ch);
    const uint32_t s0i = rotr32_dev(a, 2) ^ rotr32_dev(a, 13), s0 = s0i ^ rotr32_dev(a, 22);
    const uint32_t ab = a & b, ac = a & c, bc = b & c, maji = ab ^ ac, maj = maji ^ bc;
    put_le4_dev(o, n, 64, s0i); put_le4_dev(o, n, 68, s0); put_le4_dev(o, n, 72, ab); put_le4_dev(o, n, 76, ac);
    put_le4_dev(o, n, 80, bc); put_le4_dev(o, n, 84, maji); put_le4_dev(o, n, 88, maj);
    put_rot_dev(o, n, 92, e, 6, false); put_rot_dev(o, n, 98, e, 11, false); put_rot_dev(o, n, 104, e, 25, false);
    put_rot_dev(o, n, 110, a, 2, false); put_rot_dev(o, n, 116, a, 13, false); put_rot_dev(o, n, 122, a, 22, false);
    const uint64_t t1w = (uint64_t)h + s1 + ch + ki + wi, t2w = (uint64_t)s0 + maj;
    const uint32_t t1 = (uint32_t)t1w, t2 = (uint32_t)t2w;
    put_wadd_dev<5>(o, n, 150, t1w);
    put_wadd_dev<2>(o, n, 128, t2w);
    put_wadd_dev<2>(o, n, 134, (uint64_t)d + t1);
    put_wadd_dev<2>(o, n, 140, (uint64_t)t1 + t2);
    o[(size_t)146 * n] = meta[8 * e_ + 3];
    o[(size_t)147 * n] = meta[8 * e_ + 5];
    o[(size_t)148 * n] = meta[8 * e_ + 6];
    o[(size_t)149 * n] = meta[8 * e_ + 4] + 4 * (uint64_t)rd;
    for (int i = 0; i < 65; i++) o[(size_t)(159 + i) * n] = i == rd;
}
__global__ __launch_bounds__(256) void k_sha_compress_sponge_trace(const uint32_t* __restrict__ hx, const uint32_t* __restrict__ w,
                                                                   const uint64_t* __restrict__ meta, size_t k, size_t n,
                                                                   gl_t* __restrict__ out) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    gl_t* o = out + r;
    if (r >= k) {
        for (int c = 0; c < ZKM_SHA_COMPRESS_SPONGE_COLS; c++) o[(size_t)c * n] = 0;
        return;
    }
    uint32_t s[8], h0[8];
#pragma unroll
    for (int q = 0; q < 8; q++) s[q] = h0[q] = hx[8 * r + q];
    for (int i = 0; i < 64; i++) sha_round_dev(s, w[64 * r + i], SHA256_K_DEV[i]);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        put_le4_dev(o, n, 4 * q, h0[q]);
        put_le4_dev(o, n, 32 + 4 * q, s[q]);
        put_wadd_dev<2>(o, n, 64 + 6 * q, (uint64_t)h0[q] + s[q]);
        o[(size_t)(112 + q) * n] = meta[8 * r + 2] + 4 * (uint64_t)q;
    }
    o[(size_t)120 * n] = meta[8 * r + 4];
    o[(size_t)121 * n] = meta[8 * r + 3];
    o[(size_t)122 * n] = meta[8 * r];
    o[(size_t)123 * n] = meta[8 * r + 1];
    o[(size_t)124 * n] = meta[8 * r + 5];
    o[(size_t)125 * n] = meta[8 * r + 6];
    o[(size_t)126 * n] = 1;
}
void zkm_launch_sha_compress_trace(zkm_ctx* c, bool sponge, const uint32_t* d_hx, const uint32_t* d_w, const uint64_t* d_meta, size_t k,
                                   size_t n, gl_t* out) {
    zkm_prof_scope ps(c, sponge ? "sha_compress_sponge_trace" : "sha_compress_trace");
    if (sponge) hipLaunchKernelGGL(k_sha_compress_sponge_trace, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_hx, d_w, d_meta, k, n, out);
    else hipLaunchKernelGGL(k_sha_compress_trace, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_hx, d_w, d_meta, k, n, out);
    ZKM_HIP_CHECK(hipGetLastError());
}
