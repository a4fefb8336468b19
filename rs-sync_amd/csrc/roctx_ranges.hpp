// roctx_ranges.hpp -- optional marker ranges around the public calls and the kernel launches ("roctx
// ranges around K1-K3"), so that `rocprofv3 --marker-trace --kernel-trace` of a CLIENT shows PreSync / Sync / sync-point
// boundaries and which launches belong to them.  Off unless RSSYNC_ROCTX=1 (read once per process); the marker library is
// resolved at run time (rocprofiler-sdk's roctx, else the legacy libroctx64) -- the product links nothing for it, and with
// the switch off a range is one predictable branch.  HIP-free: included by the host solver and by the kernels' launchers.
#pragma once

#include <dlfcn.h>

#include <cstdlib>

namespace rs {

struct RoctxApi {
    using push_fn = int (*)(const char*);
    using pop_fn = int (*)();
    push_fn push = nullptr;
    pop_fn pop = nullptr;
    RoctxApi() {
        const char* e = std::getenv("RSSYNC_ROCTX");
        if (!e || !e[0] || e[0] == '0') return;
        const char* names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
        for (const char* n : names) {
            void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = (push_fn)dlsym(h, "roctxRangePushA");
            pop = (pop_fn)dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
inline const RoctxApi& roctx_api() {
    static const RoctxApi api;
    return api;
}
// RAII range on the calling thread (ranges nest: a public call's range holds its launches' ranges)
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(roctx_api().push != nullptr) {
        if (on) (void)roctx_api().push(name);
    }
    ~RoctxRange() {
        if (on) (void)roctx_api().pop();
    }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};
inline bool roctx_enabled() { return roctx_api().push != nullptr; }

} // namespace rs
