// Microbenchmark (diagnostic): the transposition between the two passes of the 16384^2 Fresnel call.  A workgroup handles ONE
// line of 16384 complex samples at a time (the line is the LDS); the intermediate image is blocked in pieces of 8 samples = 64 B.
//   layout A (shipped)  [x/8][y][x%8]: pass 1 (line y) STORES whole 64-B pieces 1 MiB apart; pass 2 (line x) LOADS 8 B of every
//                                       piece of a contiguous 1 MiB region
//   layout B            [y/8][x][y%8]: pass 1 STORES 8 B into every piece of a contiguous 1 MiB region; pass 2 LOADS whole 64-B
//                                       pieces 1 MiB apart
// 256 persistent workgroups of 256 threads (the loader waves' shape), lines handed out as the line kernels do (XCD-contiguous
// chunks, neighbouring lines on neighbouring CUs of one XCD at the same time).  Prints microseconds per line and workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int N = 16384, IB = 8;
__device__ __forceinline__ void share(int nwork, int &cstart, int &clen, int &slot, int &nslot) {
    const int xcd = blockIdx.x & 7;
    slot = blockIdx.x >> 3; nslot = gridDim.x >> 3;
    const int cq = nwork >> 3, cr = nwork & 7;
    cstart = xcd * cq + (xcd < cr ? xcd : cr); clen = cq + (xcd < cr ? 1 : 0);
}
// MODE 0: load 8 B of every piece (A pass 2); 1: load whole pieces 1 MiB apart (B pass 2); 2: store whole pieces 1 MiB apart (A pass 1);
// 3: store 8 B into every piece (B pass 1)
template <int MODE>
__global__ __launch_bounds__(256) void k(float2 *img, float *sink, int nlines) {
    int cstart, clen, slot, nslot;
    share(nlines, cstart, clen, slot, nslot);
    float acc = 0.f;
    for (int u = slot; u < clen; u += nslot) {
        const int l = cstart + u;
        if (MODE == 0 || MODE == 3) {           // element s of line l: ((l / IB) * N + s) * IB + l % IB
            float2 *base = img + ((size_t)(l / IB) * N) * IB + l % IB;
            for (int s = threadIdx.x; s < N; s += 256) {
                if (MODE == 0) { const float2 v = base[(size_t)s * IB]; acc += v.x + v.y; }
                else base[(size_t)s * IB] = make_float2((float)s, (float)l);
            }
        } else {                                // samples 8 p .. 8 p + 7 of line l: piece ((p * N) + l) * IB
            float4 *base = reinterpret_cast<float4 *>(img + (size_t)l * IB);
            for (int q = threadIdx.x; q < N / 2; q += 256) {          // 16 bytes = 2 samples per thread and step
                const int p = q >> 2, h = q & 3;
                float4 *a = base + (size_t)p * N * IB / 2 + h;
                if (MODE == 1) { const float4 v = *a; acc += v.x + v.y + v.z + v.w; }
                else *a = make_float4((float)q, (float)l, 1.f, 2.f);
            }
        }
    }
    if (acc == 12345.678f) sink[blockIdx.x] = acc;
}
// MODE 0 again through a buffer descriptor with cache-policy bits (AUX: 1 sc0, 2 nt, 16 sc1), 8-byte loads; W16: 16-byte loads (two
// neighbouring lines' samples, one of them unused) of the same pieces
template <int AUX, bool W16>
__global__ __launch_bounds__(256) void kb(float2 *img, float *sink, int nlines) {
    int cstart, clen, slot, nslot;
    share(nlines, cstart, clen, slot, nslot);
    float acc = 0.f;
    for (int u = slot; u < clen; u += nslot) {
        const int l = W16 ? ((cstart + u) & ~1) : cstart + u;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(img + ((size_t)(l / IB) * N) * IB + l % IB), 0, N * IB * 8, 0x00020000);
        for (int s = threadIdx.x; s < N; s += 256) {
            if (W16) {
                typedef int v4i __attribute__((ext_vector_type(4)));
                const v4i r = __builtin_amdgcn_raw_buffer_load_b128(rs, s * IB * 8, 0, AUX);
                acc += __int_as_float(r.x) + __int_as_float(r.y);
            } else {
                typedef int v2i __attribute__((ext_vector_type(2)));
                const v2i r = __builtin_amdgcn_raw_buffer_load_b64(rs, s * IB * 8, 0, AUX);
                acc += __int_as_float(r.x) + __int_as_float(r.y);
            }
        }
    }
    if (acc == 12345.678f) sink[blockIdx.x] = acc;
}
// layout A with pieces of IBX samples (IBX * 8 bytes): pass 2 loads 8 B of every piece, pass 1 stores whole pieces N * IBX * 8 bytes apart
template <int IBX, bool STORE>
__global__ __launch_bounds__(256) void kib(float2 *img, float *sink, int nlines) {
    int cstart, clen, slot, nslot;
    share(nlines, cstart, clen, slot, nslot);
    float acc = 0.f;
    for (int u = slot; u < clen; u += nslot) {
        const int l = cstart + u;
        if (!STORE) {
            const float2 *base = img + ((size_t)(l / IBX) * N) * IBX + l % IBX;
            for (int s = threadIdx.x; s < N; s += 256) { const float2 v = base[(size_t)s * IBX]; acc += v.x + v.y; }
        } else {                                // sample i of line l: ((i / IBX) * N + l) * IBX + i % IBX -- consecutive lanes, consecutive samples
            for (int i = threadIdx.x; i < N; i += 256) img[((size_t)(i / IBX) * N + l) * IBX + i % IBX] = make_float2((float)i, (float)l);
        }
    }
    if (acc == 12345.678f) sink[blockIdx.x] = acc;
}
int main() {
    float2 *img; float *o;
    hipMalloc(&img, sizeof(float2) * (size_t)N * N); hipMalloc(&o, 1 << 16);
    hipMemset(img, 0, sizeof(float2) * (size_t)N * N);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nlines = N;
    auto run = [&](const char *nm, auto f) {
        f(); hipDeviceSynchronize(); hipEventRecord(e0); for (int r = 0; r < 3; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("%-62s %7.3f ms per image  %6.2f us per line and workgroup  %5.2f TB/s of useful bytes\n", nm, ms, ms * 1e3 / (nlines / 256.0), 2.147 / ms);
    };
    run("A pass 2: loads, 8 B of every 64-B piece", [&] { k<0><<<256, 256>>>(img, o, nlines); });
    run("A pass 2 through a descriptor, aux 0", [&] { kb<0, false><<<256, 256>>>(img, o, nlines); });
    run("A pass 2 through a descriptor, sc0", [&] { kb<1, false><<<256, 256>>>(img, o, nlines); });
    run("A pass 2 through a descriptor, nt", [&] { kb<2, false><<<256, 256>>>(img, o, nlines); });
    run("A pass 2 through a descriptor, sc1", [&] { kb<16, false><<<256, 256>>>(img, o, nlines); });
    run("A pass 2 through a descriptor, sc0 sc1", [&] { kb<17, false><<<256, 256>>>(img, o, nlines); });
    run("A pass 2 through a descriptor, sc0 sc1 nt", [&] { kb<19, false><<<256, 256>>>(img, o, nlines); });
    run("A pass 2, 16-byte loads (half unused)", [&] { kb<0, true><<<256, 256>>>(img, o, nlines); });
    run("pieces of 4 samples (32 B): pass 2 loads 8 B of each", [&] { kib<4, false><<<256, 256>>>(img, o, nlines); });
    run("pieces of 4 samples: pass 1 stores whole pieces (8-B lanes)", [&] { kib<4, true><<<256, 256>>>(img, o, nlines); });
    run("pieces of 2 samples (16 B): pass 2 loads 8 B of each", [&] { kib<2, false><<<256, 256>>>(img, o, nlines); });
    run("pieces of 2 samples: pass 1 stores whole pieces (8-B lanes)", [&] { kib<2, true><<<256, 256>>>(img, o, nlines); });
    run("pieces of 8 samples: pass 1 stores whole pieces (8-B lanes)", [&] { kib<8, true><<<256, 256>>>(img, o, nlines); });
    run("pieces of 16 samples (128 B): pass 2 loads 8 B of each", [&] { kib<16, false><<<256, 256>>>(img, o, nlines); });
    run("pieces of 16 samples: pass 1 stores whole pieces (8-B lanes)", [&] { kib<16, true><<<256, 256>>>(img, o, nlines); });
    run("B pass 2: loads, whole 64-B pieces 1 MiB apart", [&] { k<1><<<256, 256>>>(img, o, nlines); });
    run("A pass 1: stores, whole 64-B pieces 1 MiB apart", [&] { k<2><<<256, 256>>>(img, o, nlines); });
    run("B pass 1: stores, 8 B into every 64-B piece", [&] { k<3><<<256, 256>>>(img, o, nlines); });
    return 0;
}
