// reads_text.hip -- the records of a FASTA / FASTQ text found ON THE DEVICE.
//
// The reference reads its reads with kseq (src/ILP_index.cpp:313-328, src/kseq.h:192-233): a byte-at-a-time state
// machine, 0.3 GB/s of text on one host core -- three orders of magnitude under the sketch kernel.  Here the raw
// (inflated) bytes of the file go to HBM as they are, chunk by chunk, and these kernels find the lines, check that the
// chunk is laid out in one of the two REGULAR ways, and gather the sequence bytes and the read offsets that
// phi_sketch_kernel<PROBE> takes:
//   FASTQ, four lines per record   '@' or '>' header / sequence (not empty, not starting with '>', '@', '+') / '+' line /
//                                  quality of the length of the sequence;
//   FASTA, wrapped or not          header lines ('>' or '@' first) / sequence lines / empty lines; no line starts with '+'.
// On such text kseq returns exactly these records (the sequence lines between two headers, concatenated).  Anything else
// -- a carriage return anywhere, a wrapped FASTQ record, a '+' line in a FASTA file, text before the first header -- is
// IRREGULAR: the kernels take nothing from the chunk and say so, and the caller runs the exact state machine of the host
// reader (phi_host.h) from that point of the stream on.  The bytes after the last whole record of a chunk (FASTQ: the lines
// of an unfinished group of four; FASTA: from the last header line on, since only the next header ends a record) are the
// carry: they are put in front of the next chunk, and what is left at the end of the stream goes through the host reader,
// so nothing here ever decides what a file's last bytes mean.
//
// All positions are offsets into one device buffer [carry | chunk], below 2^32.  Grids are sized by what the host knows
// (bytes, line capacity); the counts the kernels find stay in a device summary that the host reads once per chunk.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "phi_kernels.h"

#define TXT_TILE 4096u          // bytes per workgroup of the line kernels: 256 lanes x 16 bytes

namespace {

// 0x80 in every byte of v that equals c
__device__ __forceinline__ uint32_t byte_eq(uint32_t v, uint32_t c4)
{
    const uint32_t x = v ^ c4;
    const uint32_t t = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    return ~(t | x | 0x7f7f7f7fu);
}

// the 16 bytes at aligned offset `at`, with the bytes outside [start, end) replaced by 0 (never a line feed)
__device__ __forceinline__ uint4 load16(const uint8_t *buf, uint32_t at, uint32_t start, uint32_t end)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (at + 16 <= start || at >= end) return v;
    v = *reinterpret_cast<const uint4 *>(buf + at);
    if (at < start || at + 16 > end) {
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t p = at + (uint32_t)j;
            if (p < start || p >= end) w[j >> 2] &= ~(0xFFu << (8 * (j & 3)));
        }
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return v;
}

__device__ __forceinline__ uint32_t lf_flags(uint32_t v) { return byte_eq(v, 0x0a0a0a0au); }

// line feeds per tile; a carriage return anywhere makes the chunk irregular
__global__ void __launch_bounds__(256) phi_text_count_kernel(PhiTextArgs A)
{
    const uint32_t a0 = A.start & ~15u;
    const uint32_t at = a0 + blockIdx.x * TXT_TILE + threadIdx.x * 16u;
    const uint4 v = load16(A.buf, at, A.start, A.end);
    uint32_t n = __popc(lf_flags(v.x)) + __popc(lf_flags(v.y)) + __popc(lf_flags(v.z)) + __popc(lf_flags(v.w));
    const uint32_t cr = byte_eq(v.x, 0x0d0d0d0du) | byte_eq(v.y, 0x0d0d0d0du) | byte_eq(v.z, 0x0d0d0d0du) | byte_eq(v.w, 0x0d0d0d0du);
    if (cr) atomicOr(&A.sum->err, PHI_TEXT_IRREGULAR_CR);
    __shared__ uint32_t s[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) A.tile_cnt[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// exclusive prefix sums of the tile counts, in place; the number of whole lines; the first line start
__global__ void __launch_bounds__(1024) phi_text_tiles_kernel(PhiTextArgs A, uint32_t n_tiles)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_tiles ? A.tile_cnt[i] : 0;
        uint32_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) s_wave[wv] = x;
        __syncthreads();
        uint32_t before = s_carry;
        for (int k = 0; k < wv; k++) before += s_wave[k];
        if (i < n_tiles) A.tile_cnt[i] = before + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        A.sum->n_nl = s_carry;
        A.ls[0] = A.start;
        if (s_carry > A.line_cap) atomicOr(&A.sum->err, PHI_TEXT_IRREGULAR_LINES);
    }
}

// ls[i + 1] = position after the i-th line feed
__global__ void __launch_bounds__(256) phi_text_lines_kernel(PhiTextArgs A)
{
    const uint32_t a0 = A.start & ~15u;
    const uint32_t at = a0 + blockIdx.x * TXT_TILE + threadIdx.x * 16u;
    const uint4 v = load16(A.buf, at, A.start, A.end);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t n = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) n += __popc(lf_flags(w[j]));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t x = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    __shared__ uint32_t s[4];
    if (lane == 63) s[wv] = x;
    __syncthreads();
    uint32_t r = A.tile_cnt[blockIdx.x] + x - n;
    for (int k = 0; k < wv; k++) r += s[k];
    if (n == 0) return;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (((w[j >> 2] >> (8 * (j & 3))) & 0xFFu) == 0x0au) {
            if (r < A.line_cap) A.ls[r + 1] = at + (uint32_t)j + 1u;
            r++;
        }
    }
}

__device__ __forceinline__ bool is_hdr(uint32_t c) { return c == '>' || c == '@'; }

// class of every whole line and what it adds to the two running sums: records begun (high half), sequence bytes (low half)
__global__ void __launch_bounds__(256) phi_text_classify_kernel(PhiTextArgs A)
{
    const uint32_t n_nl = min(A.sum->n_nl, A.line_cap);
    const uint32_t n_val = A.mode == 1 ? (n_nl & ~3u) : n_nl;       // FASTQ: whole groups of four lines only
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_nl; i += gridDim.x * blockDim.x) {
        uint64_t add = 0;
        if (i < n_val) {
            const uint32_t b = A.ls[i], len = A.ls[i + 1] - 1u - b;
            const uint32_t c = len ? A.buf[b] : 0u;
            bool ok = true;
            if (A.mode == 1) {
                const uint32_t r = i & 3u;
                if (r == 0) { ok = len >= 1 && is_hdr(c); add = 1ull << 32; }
                else if (r == 1) { ok = len >= 1 && !is_hdr(c) && c != '+'; add = len; }
                else if (r == 2) ok = len >= 1 && c == '+';
                else ok = len == A.ls[i - 1] - 1u - A.ls[i - 2];
            } else {
                if (len == 0) add = 0;
                else if (is_hdr(c)) add = 1ull << 32;
                else if (c == '+') ok = false;
                else add = len;
                if (i == 0 && !(len >= 1 && is_hdr(c))) ok = false;  // the stream (and every carry) starts with a header line
            }
            if (!ok) { atomicOr(&A.sum->err, PHI_TEXT_IRREGULAR_LAYOUT); atomicMin(&A.sum->first_bad, i); }
        }
        A.pre[i] = add;
    }
}

// ---- exclusive scan of pre[0 .. n_nl) in place, pre[n_nl] = total: block sums, their scan, the blocks again
#define SCAN_ITEMS 8
#define SCAN_BLOCK (256 * SCAN_ITEMS)

__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t x, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t y = __shfl_up((unsigned long long)x, o, 64);
        if (lane >= o) x += y;
    }
    return x;
}

__global__ void __launch_bounds__(256) phi_text_scan_sums_kernel(PhiTextArgs A)
{
    const uint32_t n = min(A.sum->n_nl, A.line_cap);
    const uint32_t base = blockIdx.x * SCAN_BLOCK;
    if (base >= n) return;
    uint64_t t = 0;
    for (uint32_t i = base + threadIdx.x; i < min(n, base + SCAN_BLOCK); i += 256) t += A.pre[i];
    const int lane = threadIdx.x & 63;
    t = wave_incl_scan64(t, lane);
    __shared__ uint64_t s[4];
    if (lane == 63) s[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) A.blk[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void __launch_bounds__(1024) phi_text_scan_blocks_kernel(PhiTextArgs A)
{
    const uint32_t n = min(A.sum->n_nl, A.line_cap);
    const uint32_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t base = 0; base < nb; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < nb ? A.blk[i] : 0;
        const uint64_t x = wave_incl_scan64(v, lane);
        if (lane == 63) s_wave[wv] = x;
        __syncthreads();
        uint64_t before = s_carry;
        for (int k = 0; k < wv; k++) before += s_wave[k];
        if (i < nb) A.blk[i] = before + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) A.pre[n] = s_carry;
}

__global__ void __launch_bounds__(256) phi_text_scan_apply_kernel(PhiTextArgs A)
{
    const uint32_t n = min(A.sum->n_nl, A.line_cap);
    const uint32_t base = blockIdx.x * SCAN_BLOCK;
    if (base >= n) return;
    // lane t holds SCAN_ITEMS consecutive items
    const uint32_t i0 = base + threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS], t = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) { v[j] = i0 + j < n ? A.pre[i0 + j] : 0; t += v[j]; }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t x = wave_incl_scan64(t, lane);
    __shared__ uint64_t s[4];
    if (lane == 63) s[wv] = x;
    __syncthreads();
    uint64_t run = A.blk[blockIdx.x] + x - t;
    for (int k = 0; k < wv; k++) run += s[k];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        if (i0 + j < n) A.pre[i0 + j] = run;
        run += v[j];
    }
}

// read offsets of the whole records, and the summary the host reads
__global__ void __launch_bounds__(256) phi_text_finish_kernel(PhiTextArgs A)
{
    const uint32_t n_nl = min(A.sum->n_nl, A.line_cap);
    uint32_t n_rec, n_cons;                 // whole records; lines they take
    if (A.mode == 1) { n_rec = n_nl >> 2; n_cons = n_rec << 2; }
    else {
        const uint32_t n_hdr = (uint32_t)(A.pre[n_nl] >> 32);
        n_rec = n_hdr ? n_hdr - 1 : 0;
        n_cons = 0;                          // found below: the last header line
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n_nl; i += gridDim.x * blockDim.x) {
        const uint64_t p = A.pre[i];
        if (A.mode == 1) {
            if ((i & 3u) == 0 && i <= n_cons) {
                A.read_off[i >> 2] = (int64_t)(uint32_t)p;
                if (i == n_cons) { A.sum->n_bases = (uint32_t)p; A.sum->cons_end = A.ls[i]; A.sum->n_cons_lines = i; A.sum->n_rec = n_rec; }
            }
        } else if (i < n_nl) {
            // a header line: the record it begins is number (headers before it)
            const uint64_t nx = A.pre[i + 1];
            if ((nx >> 32) != (p >> 32)) {
                const uint32_t j = (uint32_t)(p >> 32);
                A.read_off[j] = (int64_t)(uint32_t)p;
                if (j == n_rec) { A.sum->n_bases = (uint32_t)p; A.sum->cons_end = A.ls[i]; A.sum->n_cons_lines = i; A.sum->n_rec = n_rec; }
            }
        } else if (n_nl == 0 || (A.pre[n_nl] >> 32) == 0) {
            // no header line among the whole lines: nothing is taken
            A.read_off[0] = 0;
            A.sum->n_bases = 0; A.sum->cons_end = A.start; A.sum->n_cons_lines = 0; A.sum->n_rec = 0;
        }
    }
}

// are the records all of one length?  (then the sketch kernel needs no offsets: phi_sketch_kernel's uniform_len)
__global__ void __launch_bounds__(256) phi_text_uniform_kernel(PhiTextArgs A)
{
    const uint32_t n_rec = A.sum->n_rec;
    if (A.sum->err || n_rec < 2) return;
    const int64_t len0 = A.read_off[1] - A.read_off[0];
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x + 1; j < n_rec; j += gridDim.x * blockDim.x)
        if (A.read_off[j + 1] - A.read_off[j] != len0) { A.sum->not_uniform = 1; return; }
}

// the sequence bytes of the records, gathered: every wave fills 2 KB of the output
#define GATHER_WAVE_BYTES 2048u
__global__ void __launch_bounds__(256) phi_text_gather_kernel(PhiTextArgs A)
{
    const uint32_t n_out = (uint32_t)A.sum->n_bases;
    const uint32_t n_lines = A.sum->n_cons_lines;
    const int lane = threadIdx.x & 63;
    uint32_t o = (blockIdx.x * 4u + (threadIdx.x >> 6)) * GATHER_WAVE_BYTES;
    if (A.sum->err || o >= n_out) return;
    const uint32_t o_end = min(n_out, o + GATHER_WAVE_BYTES);
    // the last line L with (sequence bytes before L) <= o
    uint32_t lo = 0, hi = n_lines;                       // pre[n_lines] = n_out > o
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint32_t)A.pre[mid] <= o) lo = mid; else hi = mid;
    }
    uint32_t L = lo;
    while (o < o_end) {
        const uint32_t b = (uint32_t)A.pre[L], e = (uint32_t)A.pre[L + 1];
        if (e <= o) { L++; continue; }                   // a line that holds no sequence byte at or after o
        const uint32_t d = o - b, take = min(e - o, o_end - o);
        const uint8_t *src = A.buf + A.ls[L] + d;
        uint8_t *dst = A.bases + o;
        for (uint32_t x = lane; x < take; x += 64) dst[x] = src[x];
        o += take;
        L++;
    }
}

__global__ void phi_text_noop_kernel() {}

}  // namespace

uint32_t phi_text_num_tiles(uint32_t start, uint32_t end)
{
    const uint32_t a0 = start & ~15u;
    return end > a0 ? (end - a0 + TXT_TILE - 1) / TXT_TILE : 0;
}

// Every kernel of one chunk, in order, on `st`.  A.sum must have been zeroed (first_bad = 0xFFFFFFFF) by the caller.
void phi_launch_reads_text(hipStream_t st, const PhiTextArgs &A)
{
    const uint32_t n_tiles = phi_text_num_tiles(A.start, A.end);
    if (n_tiles == 0) return;
    hipLaunchKernelGGL(phi_text_count_kernel, dim3(n_tiles), dim3(256), 0, st, A);
    hipLaunchKernelGGL(phi_text_tiles_kernel, dim3(1), dim3(1024), 0, st, A, n_tiles);
    hipLaunchKernelGGL(phi_text_lines_kernel, dim3(n_tiles), dim3(256), 0, st, A);
    const uint32_t line_blocks = std::min<uint32_t>((A.line_cap + 255) / 256, 4096);
    hipLaunchKernelGGL(phi_text_classify_kernel, dim3(line_blocks), dim3(256), 0, st, A);
    const uint32_t scan_blocks = (A.line_cap + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(phi_text_scan_sums_kernel, dim3(scan_blocks), dim3(256), 0, st, A);
    hipLaunchKernelGGL(phi_text_scan_blocks_kernel, dim3(1), dim3(1024), 0, st, A);
    hipLaunchKernelGGL(phi_text_scan_apply_kernel, dim3(scan_blocks), dim3(256), 0, st, A);
    hipLaunchKernelGGL(phi_text_finish_kernel, dim3(line_blocks), dim3(256), 0, st, A);
    hipLaunchKernelGGL(phi_text_uniform_kernel, dim3(std::min<uint32_t>(line_blocks, 1024)), dim3(256), 0, st, A);
    const uint32_t n_bytes = A.end - A.start;
    const uint32_t gather_blocks = (n_bytes + 4 * GATHER_WAVE_BYTES - 1) / (4 * GATHER_WAVE_BYTES);
    hipLaunchKernelGGL(phi_text_gather_kernel, dim3(gather_blocks), dim3(256), 0, st, A);
}

uint32_t phi_text_scan_blocks(uint32_t line_cap) { return (line_cap + SCAN_BLOCK - 1) / SCAN_BLOCK; }

void phi_warm_reads_text(hipStream_t st) { hipLaunchKernelGGL(phi_text_noop_kernel, dim3(1), dim3(64), 0, st); }
