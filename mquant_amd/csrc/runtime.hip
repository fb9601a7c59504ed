// runtime.hip -- error plumbing, version / device queries of the C ABI.
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "mq_common.h"

namespace mq {

char *last_error_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: launch failed: %s", what, hipGetErrorString(e));
    return MQ_OK;
}

// Kernels that need more than 64 KiB of dynamic LDS must be told so once PER DEVICE (the attribute
// lives in the per-device function object): with the reference's device_map="auto" placement one
// process drives several GPUs ("every buffer follows x.device").
int ensure_dynamic_lds(const void *kernel, int bytes)
{
    struct Entry { int dev; const void *fn; int granted; };
    static std::mutex mu;
    static std::vector<Entry> table;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail((int)e, "hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(mu);
    for (auto &t : table)
        if (t.dev == dev && t.fn == kernel) {
            if (t.granted >= bytes) return MQ_OK;
            e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return fail((int)e, "set dynamic LDS size %d: %s", bytes, hipGetErrorString(e));
            t.granted = bytes;
            return MQ_OK;
        }
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail((int)e, "set dynamic LDS size %d: %s", bytes, hipGetErrorString(e));
    table.push_back({dev, kernel, bytes});
    return MQ_OK;
}

// Persistent launches (gemm_pp, hadamard) size their grids by the CU count: rounded DOWN to a multiple of 8 (work ids are dealt
// to the XCDs by id % 8), never below 8; a failed query answers 256 (MI355X).
int device_cu_count()
{
    struct Entry { int dev; int cus; };
    static std::mutex mu;
    static std::vector<Entry> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(mu);
    for (auto &t : table)
        if (t.dev == dev) return t.cus;
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8) cus = prop.multiProcessorCount / 8 * 8;
    table.push_back({dev, cus});
    return cus;
}

}  // namespace mq

extern "C" int mq_version(void) { return 100; }  // 0.1.0

extern "C" const char *mq_last_error(void) { return mq::last_error_buf(); }

extern "C" int mq_device_info(char *name_host, size_t name_len, int *cu_count_host)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return mq::fail((int)e, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return mq::fail((int)e, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (name_host && name_len) snprintf(name_host, name_len, "%s (%s)", p.name, p.gcnArchName);
    if (cu_count_host) *cu_count_host = p.multiProcessorCount;
    return MQ_OK;
}
