// rssync.h -- public C++ surface of librssync_core.so (MI355X build).
//
// This is the drop-in boundary: the class below has the same virtual
// functions, in the same order, with the same signatures as the reference's
// src/core/public/rssync.h:9-29, and the factory has the same C++-linkage
// symbol (src/core/public/rssync.h:31, `_Z17CreateSyncProblemv`), so a client
// built against the reference header binds to this library unchanged.  The
// destructor is defined out of line in the library (reference:
// src/core/core_private.cpp:365), which is where the vtable and typeinfo live.
//
// Units: seconds everywhere except `timestamps_us` (microseconds).
// Ownership: every pointer argument is borrowed for the duration of the call;
// the library copies what it keeps (reference contract: SURVEY.md 8(b)).
// Errors: none are returned.  Invalid input or a device failure writes the
// reason to ./panic.txt and terminates the process with status 1, like the
// reference's panic_to_file (src/core_support/panic.cpp:7-15).
// Threading: one object, one thread at a time; calls block until done.
#pragma once

#include <cstddef>
#include <cstdint>
#include <utility>

#if defined(_WIN32)
#if RSSYNC_EXPORTS
#define RSSYNC_API __declspec(dllexport)
#else
#define RSSYNC_API __declspec(dllimport)
#endif
#else
#define RSSYNC_API
#endif

class ISyncProblem {
   public:
    virtual ~ISyncProblem();

    // Gyro orientation samples at a fixed rate: `count` consecutive [w,x,y,z]
    // quaternions, sample i taken at first_timestamp + i / sample_rate.
    virtual void SetGyroQuaternions(const double* data, size_t count, double sample_rate,
                                    double first_timestamp) = 0;
    // Gyro orientation samples with their own ascending timestamps; resampled
    // (slerp) onto a fixed grid whose rate is the measured rate rounded to 50 Hz.
    virtual void SetGyroQuaternions(const int64_t* timestamps_us, const double* quats,
                                    size_t count) = 0;
    // Feature tracks between `frame` and the next frame: unit rays (xyz
    // interleaved, lens already undistorted) and the capture time of each ray's
    // image row.  Setting a frame again replaces it.
    // Any count, as in the reference (core_private.cpp:192-203), up to the indexing bound
    // of 2^24 tracks per frame (rship_max_tracks() in rssync_hip.h).  A frame's kernels follow
    // from ITS OWN track count (size classes): up to 512 tracks one wave per
    // frame, up to 8192 a workgroup of four waves (PreSync's sweep of a frame of more than 6144
    // tracks: eight); only the frames of more than 8192 tracks themselves run the
    // slower, exact variants of the kernels -- a correctness path, not a tuned one -- whatever
    // else the problem holds, and a frame's results do not depend on its neighbours.
    virtual void SetTrackResult(int64_t frame, const double* ts_a, const double* ts_b,
                                const double* rays_a, const double* rays_b, size_t count) = 0;
    // Brute-force sweep of the delay over initial_delay +- search_radius in
    // search_step increments on frames [frame_begin, frame_end).  -> {cost, delay}
    virtual std::pair<double, double> PreSync(double initial_delay, int64_t frame_begin,
                                              int64_t frame_end, double search_step,
                                              double search_radius) = 0;
    // Non-linear refinement on frames [frame_begin, frame_end] (end inclusive);
    // stops early when the delay leaves search_center +- search_radius.  -> {cost, delay}
    virtual std::pair<double, double> Sync(double initial_delay, int64_t frame_begin,
                                           int64_t frame_end, double search_center,
                                           double search_radius) = 0;

    // The PreSync cost on point_count evenly spaced delays (both ends included),
    // written to the caller's arrays.
    virtual void DebugPreSync(double initial_delay, int64_t frame_begin, int64_t frame_end,
                              double search_radius, double* delays, double* costs,
                              int point_count) = 0;
};

RSSYNC_API ISyncProblem* CreateSyncProblem();
