// betaone_amd/csrc/bo_rt.h -- the few HIP runtime calls the engine's host side makes.
// (With BO_WAVE_EMU -- tests/wave_emulator only -- they are mapped onto the CPU wave emulator.)
#pragma once
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#if defined(BO_WAVE_EMU)
static inline int rt_set_device(int) { return 0; }
static inline int rt_malloc(void **p, size_t n) { *p = malloc(n ? n : 1); if (*p) memset(*p, 0xCD, n); return *p ? 0 : -1; }
static inline void rt_free(void *p) { free(p); }
static inline int rt_h2d(void *d, const void *h, size_t n, void *) { memcpy(d, h, n); return 0; }
static inline int rt_d2h(void *h, const void *d, size_t n, void *) { memcpy(h, d, n); return 0; }
static inline int rt_memset(void *d, int v, size_t n, void *) { memset(d, v, n); return 0; }
static inline int rt_sync(void *) { return 0; }
static inline int rt_host_alloc(void **p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? 0 : -1; }
static inline void rt_host_free(void *p) { free(p); }
static inline int rt_h2d_2d(void *d, size_t dpitch, const void *h, size_t hpitch, size_t width, size_t height, void *) {
    for (size_t r = 0; r < height; r++) memcpy((char *)d + r * dpitch, (const char *)h + r * hpitch, width);
    return 0;
}
static inline int rt_d2h_2d(void *h, size_t hpitch, const void *d, size_t dpitch, size_t width, size_t height, void *) {
    for (size_t r = 0; r < height; r++) memcpy((char *)h + r * hpitch, (const char *)d + r * dpitch, width);
    return 0;
}
typedef int rt_event;  // (copies are synchronous on the emulator: an event has nothing to wait for)
static inline int rt_event_create(rt_event *) { return 0; }
static inline int rt_event_record(rt_event, void *) { return 0; }
static inline int rt_event_sync(rt_event) { return 0; }
static inline void rt_event_destroy(rt_event) {}
static inline const char *rt_errstr(int) { return "emulator error"; }
#define RT_LAUNCH(kernel, grid, stream, ...) (bo_emu::launch((grid), [&]() { kernel(__VA_ARGS__); }), 0)
#else
#include <hip/hip_runtime.h>
static inline int rt_set_device(int d) { return (int)hipSetDevice(d); }
static inline int rt_malloc(void **p, size_t n) { return (int)hipMalloc(p, n ? n : 1); }
static inline void rt_free(void *p) { (void)hipFree(p); }
static inline int rt_h2d(void *d, const void *h, size_t n, void *s) { return (int)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)s); }
static inline int rt_d2h(void *h, const void *d, size_t n, void *s) { return (int)hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)s); }
static inline int rt_memset(void *d, int v, size_t n, void *s) { return (int)hipMemsetAsync(d, v, n, (hipStream_t)s); }
static inline int rt_sync(void *s) { return (int)hipStreamSynchronize((hipStream_t)s); }
// pinned host staging: async copies to/from it are one DMA without the runtime's pageable bounce buffer
static inline int rt_host_alloc(void **p, size_t n) { return (int)hipHostMalloc(p, n ? n : 1, hipHostMallocDefault); }
static inline void rt_host_free(void *p) { (void)hipHostFree(p); }
static inline int rt_h2d_2d(void *d, size_t dpitch, const void *h, size_t hpitch, size_t width, size_t height, void *s) {
    return (int)hipMemcpy2DAsync(d, dpitch, h, hpitch, width, height, hipMemcpyHostToDevice, (hipStream_t)s);
}
static inline int rt_d2h_2d(void *h, size_t hpitch, const void *d, size_t dpitch, size_t width, size_t height, void *s) {
    return (int)hipMemcpy2DAsync(h, hpitch, d, dpitch, width, height, hipMemcpyDeviceToHost, (hipStream_t)s);
}
typedef hipEvent_t rt_event;
static inline int rt_event_create(rt_event *ev) { return (int)hipEventCreateWithFlags(ev, hipEventDisableTiming); }
static inline int rt_event_record(rt_event ev, void *s) { return (int)hipEventRecord(ev, (hipStream_t)s); }
static inline int rt_event_sync(rt_event ev) { return (int)hipEventSynchronize(ev); }
static inline void rt_event_destroy(rt_event ev) { (void)hipEventDestroy(ev); }
static inline const char *rt_errstr(int e) { return hipGetErrorString((hipError_t)e); }
#define RT_LAUNCH(kernel, grid, stream, ...)                                                                   \
    ([&]() -> int {                                                                                            \
        hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3(64), 0, (hipStream_t)(stream), __VA_ARGS__);   \
        return (int)hipGetLastError();                                                                         \
    }())
#endif
