/* hip_stub.h - controls of the device-free HIP stand-in (tests/hipstub/hip_stub.cpp; TEST INFRASTRUCTURE, never shipped):
 * fault injection and accounting for the host-side tests of libfx8010_amd.so under sanitizers. */
#ifndef FX_HIP_STUB_H
#define FX_HIP_STUB_H
#ifdef __cplusplus
extern "C" {
#endif
/* the next `count` calls from the from_nth-th one on (0 = the very next) fail; from_nth < 0 switches the fault off */
void fxstub_fail_module_loads(long from_nth, long count);                 /* hipModuleLoadData -> hipErrorSharedObjectInitFailed */
void fxstub_fail_mallocs(long from_nth, long count);                      /* hipMalloc / hipHostMalloc -> hipErrorOutOfMemory */
void fxstub_fail_launches(long from_nth, long count, int hip_error);      /* hipModuleLaunchKernel -> hip_error */
void fxstub_set_capacity(unsigned long long bytes);                       /* "device memory": allocations beyond it fail with out-of-memory */
void fxstub_set_kernel_micros(int us);                                    /* how long a stand-in kernel takes */
unsigned long long fxstub_bytes_in_use(void);
long fxstub_live_allocations(void);
long fxstub_module_loads(void);
long fxstub_live_modules(void);
long fxstub_kernels_run(void);
long fxstub_bad_pcm_launches(void);        /* launches whose PCM buffers were not wholly device-addressable, or overlapped without being one buffer */
long fxstub_cross_device_errors(void);     /* launches that mixed devices: a module, a stream, the current device or a buffer of another device */
#ifdef __cplusplus
}
#endif
#endif
