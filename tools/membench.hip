// tools/membench.hip — random-line gather microbenchmark (dev tool): what is the MI355X ceiling for dependent random
// 12-byte reads, and is an HBM fill 64 B or 128 B?   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// MODE 0: one 12-B read per step at a random 64-B line
// MODE 1: two 12-B reads per step, both halves of one random 128-B line (independent)
// MODE 2: two 12-B reads per step at two independent random 64-B lines
// MODE 3: one 12-B read per step, CHAINS independent chains per lane (ILP)
// MODE 4: the whole random 64-B line per step as 4 x dwordx4 (what a DFS node does)
// MODE 5: one 8-B read per step
// MODE 6: 64-B line as 4 x dwordx4, but 4 adjacent lanes share a line (quad-cooperative)
// MODE 7: a random 32-B sector per step as 2 x dwordx4 (Format D: 64 rows in 32 bytes)
// MODE 8: a random 128-B line per step as dwordx4 @0, dwordx4 @16, dwordx2 @32, dwordx3 @40 + 12 k (Format S)
// MODE 9: a random 128-B line per step as dword @4 k, 4 x dwordx4 @64 (Format P)
template <int MODE, int CHAINS>
__global__ __launch_bounds__(256) void gather(const uint8_t* __restrict__ buf, uint64_t nlines, int steps, uint64_t* __restrict__ out) {
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t st[CHAINS];
    for (int c = 0; c < CHAINS; ++c) st[c] = mix(gid * CHAINS + c + 1);
    uint64_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        for (int c = 0; c < CHAINS; ++c) {
            uint64_t line = st[c] % nlines;
            if (MODE == 1) line &= ~1ull;
            const uint32_t* p = reinterpret_cast<const uint32_t*>(buf + line * 64 + (st[c] >> 60) % 5 * 12);
            uint64_t v = (uint64_t)p[0] + p[1] + p[2];
            if (MODE == 4) { const uint4* b4 = reinterpret_cast<const uint4*>(buf + line * 64); v = 0; for (int k = 0; k < 4; ++k) { uint4 t = b4[k]; v += (uint64_t)t.x + t.y + t.z + t.w; } }
            if (MODE == 7) { const uint4* b4 = reinterpret_cast<const uint4*>(buf + (st[c] % (nlines * 2)) * 32); uint4 t = b4[0], u = b4[1]; v = (uint64_t)t.x + t.y + t.z + t.w + u.x + u.y + u.z + u.w; }
            if (MODE == 8) { const uint8_t* L = buf + (st[c] % (nlines / 2)) * 128; uint4 t = *reinterpret_cast<const uint4*>(L), u = *reinterpret_cast<const uint4*>(L + 16); uint2 w = *reinterpret_cast<const uint2*>(L + 32);
                             const uint32_t* g = reinterpret_cast<const uint32_t*>(L + 40 + 12 * ((st[c] >> 58) % 7)); v = (uint64_t)t.x + t.y + t.z + t.w + u.x + u.y + u.z + u.w + w.x + w.y + g[0] + g[1] + g[2]; }
            if (MODE == 9) { const uint8_t* L = buf + (st[c] % (nlines / 2)) * 128; v = reinterpret_cast<const uint32_t*>(L)[(st[c] >> 58) & 15]; const uint4* b4 = reinterpret_cast<const uint4*>(L + 64); for (int k = 0; k < 4; ++k) { uint4 t = b4[k]; v += (uint64_t)t.x + t.y + t.z + t.w; } }
            if (MODE == 5) { v = *reinterpret_cast<const uint64_t*>(buf + line * 64 + ((st[c] >> 60) & 7) * 8); }
            if (MODE == 6) {   // lanes 4j..4j+3 fetch the four 16-B pieces of the lines of lanes 4j..4j+3 in turn
                v = 0;
                for (int k = 0; k < 4; ++k) {
                    uint64_t l2 = __shfl(line, (threadIdx.x & ~3u) + k, 64);
                    uint4 t = *reinterpret_cast<const uint4*>(buf + l2 * 64 + (threadIdx.x & 3u) * 16);
                    // hand the piece to its owner: sum over the quad stands in for the transpose
                    uint32_t x = t.x + t.y + t.z + t.w;
                    x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64);
                    if ((threadIdx.x & 3u) == (unsigned)k) v = x;
                }
            }
            if (MODE == 1) { const uint32_t* q = p + 16; v += (uint64_t)q[0] + q[1] + q[2]; }
            if (MODE == 2) { uint64_t l2 = mix(st[c]) % nlines; const uint32_t* q = reinterpret_cast<const uint32_t*>(buf + l2 * 64); v += (uint64_t)q[0] + q[1] + q[2]; }
            acc += v;
            st[c] = mix(st[c] + v);       // next address depends on the data
        }
    }
    out[gid] = acc;
}

template <int MODE, int CHAINS>
double run(const uint8_t* buf, uint64_t nlines, int steps, uint64_t* out, int blocks) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    gather<MODE, CHAINS><<<blocks, 256>>>(buf, nlines, 8, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    gather<MODE, CHAINS><<<blocks, 256>>>(buf, nlines, steps, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double lane_steps = (double)blocks * 256 * steps * CHAINS;
    printf("mode %d chains %d blocks %d: %.2f ms, %.2f G steps/s\n", MODE, CHAINS, blocks, ms, lane_steps / ms / 1e6);
    return ms;
}

int main(int argc, char** argv) {
    uint64_t bytes = argc > 1 ? strtoull(argv[1], 0, 10) : (3ull << 30);
    uint64_t nlines = bytes / 64;
    uint8_t* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    uint64_t* out; CK(hipMalloc(&out, 8ull * 8192 * 256));
    int steps = 400;
    printf("buffer %.2f GB\n", bytes / 1e9);
    bool quick = argc > 2;
    for (int blocks : {2048}) {
        run<0, 1>(buf, nlines, steps, out, blocks);
        if (quick) { run<7, 1>(buf, nlines, steps, out, blocks); run<8, 1>(buf, nlines, steps, out, blocks); run<9, 1>(buf, nlines, steps, out, blocks); run<4, 1>(buf, nlines, steps, out, blocks); continue; }
        run<1, 1>(buf, nlines, steps, out, blocks);
        run<2, 1>(buf, nlines, steps, out, blocks);
        run<3, 2>(buf, nlines, steps, out, blocks);
        run<3, 4>(buf, nlines, steps, out, blocks);
        run<4, 1>(buf, nlines, steps, out, blocks);
        run<5, 1>(buf, nlines, steps, out, blocks);
        run<6, 1>(buf, nlines, steps, out, blocks);
    }
    return 0;
}
