// sync_driver.cpp -- the reference driver's call pattern (src/core_testcode.cpp:248,52,157,303-316)
// through the C++ ISyncProblem surface, on inputs read from a flat binary file instead of a video
// (the video/optical-flow/telemetry stages of the reference driver are out of scope).
//
//   g++ -std=c++17 -Iinclude examples/sync_driver.cpp -Lrs-sync_amd -lrssync_core \
//       -Wl,-rpath,$PWD/rs-sync_amd -Wl,-rpath,/opt/rocm/lib -o sync_driver
//   ./sync_driver input.bin            -> prints "pos,delay_ms" per sync point, like the reference's CSV
//   ./sync_driver input.bin batched    -> same output from ONE rssync_ext_sync_points call
//
// File layout (little endian): int64 n_gyro, double sample_rate, double first_timestamp,
// n_gyro x 4 doubles [w,x,y,z]; int64 n_frames; per frame: int64 id, int64 n, ts_a[n], ts_b[n],
// rays_a[3n], rays_b[3n] (doubles); then int64 window, int64 distance, double initial_ms,
// double step_ms, double radius_ms.
#include "rssync.h"
#include "rssync_c.h"

#include <cstdint>
#include <cstdio>
#include <memory>
#include <vector>

template <typename T>
static bool rd(FILE* f, T* p, size_t n = 1) { return fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::unique_ptr<ISyncProblem> sp(CreateSyncProblem()); // core_testcode.cpp:248
    int64_t n_gyro = 0, n_frames = 0;
    double fs = 0, t0 = 0;
    rd(f, &n_gyro); rd(f, &fs); rd(f, &t0);
    std::vector<double> q((size_t)n_gyro * 4);
    rd(f, q.data(), q.size());
    sp->SetGyroQuaternions(q.data(), (size_t)n_gyro, fs, t0);
    rd(f, &n_frames);
    int64_t first = 0, last = 0;
    for (int64_t i = 0; i < n_frames; ++i) {
        int64_t id = 0, n = 0;
        rd(f, &id); rd(f, &n);
        std::vector<double> ta(n), tb(n), ra(3 * n), rb(3 * n);
        rd(f, ta.data(), ta.size()); rd(f, tb.data(), tb.size());
        rd(f, ra.data(), ra.size()); rd(f, rb.data(), rb.size());
        sp->SetTrackResult(id, ta.data(), tb.data(), ra.data(), rb.data(), (size_t)n); // :157, buffers freed right after
        if (i == 0) first = id;
        last = id;
    }
    int64_t window = 0, distance = 0;
    double initial_ms = 0, step_ms = 0, radius_ms = 0;
    rd(f, &window); rd(f, &distance); rd(f, &initial_ms); rd(f, &step_ms); rd(f, &radius_ms);
    fclose(f);
    if (argc > 2) { // the same loop as one batched call on the object we already hold
        std::vector<int64_t> positions;
        for (int64_t pos = first; pos + window < last + 1; pos += distance) positions.push_back(pos);
        std::vector<double> delays(positions.size());
        rssync_problem* h = rssync_ext_borrow(sp.get());
        if (!h || rssync_ext_sync_points(h, positions.data(), (int)positions.size(), window, initial_ms / 1000, 1,
                                         step_ms / 1000., radius_ms / 1000., 4, nullptr, delays.data()))
            return 3;
        rssync_destroy(h); // the handle only; sp still owns the problem
        for (size_t i = 0; i < positions.size(); ++i) std::printf("%lld,%.9f\n", (long long)positions[i], 1000 * delays[i]);
        return 0;
    }
    for (int64_t pos = first; pos + window < last + 1; pos += distance) { // :270-274
        const double initial = initial_ms / 1000;
        double delay = sp->PreSync(initial, pos, pos + window, step_ms / 1000., radius_ms / 1000.).second; // :310
        for (int i = 0; i < 4; ++i) delay = sp->Sync(delay, pos, pos + window, initial, radius_ms / 1000.).second; // :314
        std::printf("%lld,%.9f\n", (long long)pos, 1000 * delay); // :315
    }
    return 0;
}
