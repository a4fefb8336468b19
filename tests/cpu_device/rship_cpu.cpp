// rship_cpu.cpp -- TEST DOUBLE for the device C-ABI (include/rssync_hip.h).
//
// The product's host solver (rs-sync_amd/csrc/sync_problem.cpp) talks to the GPU through the
// rship_* entry points.  To exercise that host code on a machine without a GPU -- the Sync
// control loop, frame selection, delay splitting, panics, and the multi-rank reduce hook under
// gloo -- the tests link the SAME sync_problem.cpp against this file instead of the HIP
// translation unit.  It evaluates the kernels' arithmetic (the shared headers device_math.hpp and
// sync_math.hpp) on the CPU.
//
// For Sync (fp64) it is also the DEVICE-ASSOCIATION ORACLE: the row terms come from the same source as the
// kernels' (contraction off on both sides), and the sums over rows are taken in the kernels' own order --
// rows-per-thread partials, the row_shr / readlane wave tree of wave_sum_f64, waves in order -- so that the
// GPU library and this stand-in must agree BIT FOR BIT on every loss, gradient, motion estimate and Sync trace
// (tests/test_gpu_bitexact.py).  What then differs between this file and oracle/rssync_oracle.c (sequential
// sums, libm's log1p) is the effect of reassociation alone, measured on the CPU (profiles/r3_reassociation.json).
// PreSync's fp32 search is evaluated sequentially here (the device uses hardware reciprocals there): the
// tests compare it statistically, and start both sides of a bit-exactness test from the same winners.  It is built only by tests/conftest.py into
// tests/_build/librssync_hosttest.so, is never part of librssync_core.so, and nothing under
// rs-sync_amd/ can load it.
#include "../../include/rssync_hip.h"
#include "../../rs-sync_amd/csrc/device_math.hpp"
#include "../../rs-sync_amd/csrc/sync_math.hpp"
#include "../../rs-sync_amd/csrc/lens_math.hpp"
#include "../../rs-sync_amd/csrc/gyro_math.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using rs::d3;
using rs::d4;
using rs::f3;
using rs::f4;

static const int32_t kNone = (int32_t)0x80000000;

struct rship_ctx {
    std::string err;
    std::vector<f4> coef; // 4 per knot (fp32, PreSync)
    std::vector<double> coef64; // 16 per knot (Sync)
    double fs = 0;
    std::vector<double> knots, g_ts, g_rates; // gyro pipeline: grid knots, uploaded rates
    std::vector<double> raw;    // mirror of the host staging arena
    std::vector<f4> rays_a, rays_b;
    std::vector<double> q[4];   // fp64 streams {ax,bx} {ay,by} {az,bz} {ta,tb}, 2 doubles per ray each
    std::vector<rship_frame> frames;
    std::vector<uint32_t> sel, grp, grp_off; // slots, slot -> window, window offsets
    std::vector<double> M, k;                // per slot
    int lbfgs_reeval = 0;
    // reduction plan and the results of the last *_enqueue
    std::vector<uint32_t> plan_idx, plan_chunk_off, plan_win_off;
    bool plan_has_idx = false;
    std::vector<double> win_out, chunk_out, frame_cost;
    std::vector<int32_t> best_h;
    uint32_t pend_rows = 0, pend_flags = 0;
    // a batch of sweeps collected together (rship_presync_batch_begin)
    uint32_t batch_n = 0, batch_rows = 0;
    std::vector<std::vector<double>> batch_win, batch_chunk;
    std::vector<uint32_t> batch_flags;
    std::vector<int32_t> batch_gyro_status; // of the orientations enqueued with rship_gyro_rates_integrate_enqueue
    bool pend_grad = false;
    std::vector<int32_t> init_h; // per slot: winning hypothesis of a pending GuessMotion, or kNone
    uint64_t init_seed = 0;
    uint32_t init_stream = 0, init_stride = 0;
};

namespace {

int fail(rship_ctx* c, const char* m) { c->err = m; return 1; }

void row(const rship_ctx* c, const rship_frame& fr, uint32_t i, int32_t kd, float fd, f3& P, f3* dP) {
    const int n = (int)(c->coef.size() / 4);
    const f4 A = c->rays_a[fr.ray_offset + i], B = c->rays_b[fr.ray_offset + i]; // {ax,bx,ay,by} {az,bz,ta,tb}
    const f4 ra{A.x, A.z, B.x, B.z}, rb{A.y, A.w, B.y, B.w};
    f3 r[2], dr[2];
    const f4 rays[2] = {ra, rb};
    for (int s = 0; s < 2; ++s) {
        rs::Knot kn = rs::spline_locate(rays[s].w, fr.base_knot + kd, fd, n);
        const f4* p = &c->coef[(size_t)kn.ci * 4];
        if (dP) rs::rotate_ray<true>(p[0], p[1], p[2], p[3], kn, f3{rays[s].x, rays[s].y, rays[s].z}, r[s], dr[s]);
        else rs::rotate_ray<false>(p[0], p[1], p[2], p[3], kn, f3{rays[s].x, rays[s].y, rays[s].z}, r[s], dr[s]);
    }
    P = rs::cross(r[0], r[1]);
    if (dP) *dP = rs::add(rs::cross(dr[0], r[1]), rs::cross(r[0], dr[1]));
}

// one row of P in fp64 from the fp64 streams and the fp64 spline table (kernels/sync64.hpp: residual_row64)
void row64(const rship_ctx* c, const rship_frame& fr, uint32_t i, int32_t kd, double fd, d3& P, d3* dP) {
    const int n = (int)(c->coef64.size() / 16);
    const size_t o = (size_t)fr.ray_offset + i;
    const d3 ra{c->q[0][2 * o], c->q[1][2 * o], c->q[2][2 * o]}, rb{c->q[0][2 * o + 1], c->q[1][2 * o + 1], c->q[2][2 * o + 1]};
    const double t[2] = {c->q[3][2 * o], c->q[3][2 * o + 1]};
    const d3 rays[2] = {ra, rb};
    d3 r[2], dr[2];
    for (int s = 0; s < 2; ++s) {
        rs::KnotT<double> kn = rs::spline_locate(t[s], fr.base_knot + kd, fd, n);
        const double* p = &c->coef64[(size_t)kn.ci * 16];
        const d4 y{p[0], p[1], p[2], p[3]}, b{p[4], p[5], p[6], p[7]}, cc{p[8], p[9], p[10], p[11]}, d{p[12], p[13], p[14], p[15]};
        if (dP) rs::rotate_ray<true>(y, b, cc, d, kn, rays[s], r[s], dr[s]);
        else rs::rotate_ray<false>(y, b, cc, d, kn, rays[s], r[s], dr[s]);
    }
    P = rs::cross(r[0], r[1]);
    if (dP) *dP = rs::add(rs::cross(dr[0], r[1]), rs::cross(r[0], dr[1]));
}

struct Rows {
    std::vector<f3> n;
    std::vector<float> nrm;
    bool bad = false;
};

Rows unit_rows(const rship_ctx* c, const rship_frame& fr, int32_t kd, float fd) {
    Rows t;
    t.n.resize(fr.n_rays);
    t.nrm.resize(fr.n_rays);
    for (uint32_t i = 0; i < fr.n_rays; ++i) {
        f3 P;
        row(c, fr, i, kd, fd, P, nullptr);
        float n2 = rs::dot(P, P);
        if (!std::isfinite(n2)) t.bad = true;
        bool tiny = n2 < 1e-24f;
        float inv = tiny ? 1.f : 1.0f / std::sqrt(n2);
        t.n[i] = rs::scale(P, inv);
        t.nrm[i] = tiny ? 1.f : n2 * inv;
    }
    return t;
}

f3 hypothesis(const Rows& t, uint64_t seed, int64_t frame, uint32_t stream, uint32_t h) {
    uint32_t i0, i1;
    rs::sample_pair(seed, frame, stream, h, (uint32_t)t.n.size(), i0, i1);
    // safe_normalize on the UN-NORMALISED cross product (core_private.cpp:45-46): P_i = nrm_i n_i
    // (The stand-in's sweep is NOT the bit-level double of the device's: its rows come from the fp64 streams and are rounded
    // once, its norms are stage A's, its search is sequential -- the device's fp32 kernels recompute the two norms with
    // row_scale_general and, on near-static pairs, take all rows from the fp64 streams (kernels/lmeds.hpp, "fp64 rows").
    // Near the 1e-12 threshold the two can decide differently in the last bits; what is bit-identical between stand-in and
    // device is Sync's fp64 arithmetic, started from the same winners -- tests/test_gpu_bitexact.py.  ADVICE r5.)
    f3 v = rs::cross(t.n[i0], t.n[i1]);
    float nn = std::sqrt(rs::dot(v, v));
    const float ss = t.nrm[i0] * t.nrm[i1];
    if (!(ss * nn < 1e-12f)) v = rs::scale(v, 1.0f / nn);
    else v = rs::scale(v, ss);
    return v;
}

// LMedS arg-min over hypotheses of the lower quartile of |r| (ties: first wins)
int lmeds(const Rows& t, uint32_t n_hyp, uint64_t seed, int64_t frame, uint32_t stream, f3& Mv) {
    const size_t N = t.n.size(), kq = N / 4;
    std::vector<float> a(N);
    float best = INFINITY;
    int bh = -1;
    for (uint32_t h = 0; h < n_hyp; ++h) {
        f3 v = hypothesis(t, seed, frame, stream, h);
        for (size_t i = 0; i < N; ++i) a[i] = std::fabs(rs::dot(t.n[i], v));
        std::nth_element(a.begin(), a.begin() + kq, a.end());
        if (a[kq] < best) { best = a[kq]; bh = (int)h; }
    }
    Mv = f3{0, 0, 0};
    if (bh >= 0) Mv = hypothesis(t, seed, frame, stream, (uint32_t)bh);
    return bh;
}

float clampk(float k) { return (k < 10.f) ? 10.f : ((1000.f < k) ? 1000.f : k); }

} // namespace

extern "C" {

int rship_max_tracks(void) { return 1 << 24; }
int rship_create(rship_ctx** out, int) { *out = new rship_ctx(); return 0; } // any device ordinal: a fake device
void rship_destroy(rship_ctx* c) { delete c; }
const char* rship_last_error(const rship_ctx* c) { return c->err.c_str(); }
int rship_set_stream(rship_ctx*, void*) { return 0; }
int rship_set_option(rship_ctx* c, int option, int value) {
    if (option == RSHIP_OPT_TRACKS_HINT) return 0; // (ignored since round 5: a frame's shape follows its own track count)
    if (option != RSHIP_OPT_LBFGS_REEVAL) return fail(c, "set_option: unknown option");
    c->lbfgs_reeval = value != 0;
    return 0;
}

// ---- gyro pipeline: the device's formulas (gyro_math.hpp), one sample after the other ----------------------
static int table_from_knots(rship_ctx* c, double sample_rate) {
    const size_t n = c->knots.size() / 4;
    if (n < 2) return fail(c, "spline: need >= 2 knots");
    rs::SplinePivots piv;
    piv.cp[0] = 0.0;
    for (int i = 1; i < rs::kSplinePivots; ++i) piv.cp[i] = rs::spline_next_pivot(piv.cp[i - 1]);
    c->coef64.assign(n * 16, 0.0);
    std::vector<double> cf(n), cc(n);
    for (int comp = 0; comp < 4; ++comp) {
        auto y = [&](size_t i) { return c->knots[4 * i + comp]; };
        cf[0] = 0.0;
        for (size_t i = 1; i + 1 < n; ++i) cf[i] = rs::spline_forward(y(i - 1), y(i), y(i + 1), rs::spline_pivot(piv, (uint32_t)(i - 1)), cf[i - 1]);
        cf[n - 1] = 0.0;
        cc[n - 1] = 0.0;
        cc[0] = 0.0;
        for (size_t i = n - 1; i-- > 1;) cc[i] = rs::spline_backward(cf[i], rs::spline_pivot(piv, (uint32_t)i), cc[i + 1]);
        double b_prev = 0, d_prev = 0;
        for (size_t i = 0; i < n; ++i) {
            double b, d;
            if (i + 1 < n) rs::spline_segment(y(i), y(i + 1), cc[i], cc[i + 1], &b, &d);
            else rs::spline_tail(b_prev, d_prev, cc[n - 2], &b, &d);
            double* r = &c->coef64[16 * i];
            r[comp] = y(i); r[4 + comp] = b; r[8 + comp] = cc[i]; r[12 + comp] = d;
            b_prev = b;
            d_prev = d;
        }
    }
    c->coef.resize(n * 4);
    float* f = reinterpret_cast<float*>(c->coef.data());
    for (size_t i = 0; i < n * 16; ++i) f[i] = (float)c->coef64[i];
    c->fs = sample_rate;
    return 0;
}

int rship_gyro_uniform(rship_ctx* c, const double* quats, uint32_t n, double sample_rate) {
    c->knots.assign(quats, quats + 4 * (size_t)n);
    return table_from_knots(c, sample_rate);
}

static int timestamped(rship_ctx* c, const int64_t* ts, const double* quats, uint32_t n, bool bad_input, rship_gyro_result* out) {
    const int grid_status = rs::grid_of(ts[0], ts[n - 1], n, rs::kMaxKnots, out);
    uint32_t out_of_order = 0;
    for (uint32_t i = 1; i < n && !out_of_order; ++i)
        if (ts[i - 1] > ts[i]) out_of_order = i;
    bool bad_knot = false;
    if (grid_status == RSHIP_GYRO_OK) {
        c->knots.resize(4 * (size_t)out->n_knots);
        for (uint32_t i = 0; i < out->n_knots; ++i)
            if (!rs::resample_knot(ts, quats, n, rs::grid_time_us(out->first_sample + i, (uint64_t)out->fs), &c->knots[4 * (size_t)i])) bad_knot = true;
        if (table_from_knots(c, out->fs)) return 1;
    }
    out->status = RSHIP_GYRO_OK;
    if (bad_input) out->status = RSHIP_GYRO_BAD_INPUT;
    else if (grid_status == RSHIP_GYRO_BAD_RATE || grid_status == RSHIP_GYRO_TOO_LARGE) out->status = grid_status;
    else if (out_of_order) {
        out->status = RSHIP_GYRO_OUT_OF_ORDER;
        out->bad_pos = out_of_order;
        out->bad_a = ts[out_of_order - 1];
        out->bad_b = ts[out_of_order];
    } else if (grid_status != RSHIP_GYRO_OK) out->status = grid_status;
    else if (bad_knot) out->status = RSHIP_GYRO_BAD_KNOT;
    else if (!std::isfinite(out->fs)) out->status = RSHIP_GYRO_BAD_RATE;
    else if (!std::isfinite(out->start)) out->status = RSHIP_GYRO_BAD_START;
    if (out->status != RSHIP_GYRO_OK) { c->knots.clear(); c->coef.clear(); c->coef64.clear(); }
    return 0;
}

int rship_gyro_timestamped(rship_ctx* c, const int64_t* ts_us, const double* quats, uint32_t n, rship_gyro_result* out) {
    if (n < 2) return fail(c, "gyro: need >= 2 samples");
    return timestamped(c, ts_us, quats, n, false, out);
}

int rship_gyro_rates_upload(rship_ctx* c, const double* ts_s, const double* rates, uint32_t n) {
    if (n < 2) return fail(c, "gyro: need >= 2 samples");
    c->g_ts.assign(ts_s, ts_s + n);
    c->g_rates.assign(rates, rates + 3 * (size_t)n);
    return 0;
}

int rship_gyro_rates_integrate(rship_ctx* c, const int32_t axis[3], const double sign[3], rship_gyro_result* out) {
    const uint32_t n = (uint32_t)c->g_ts.size();
    if (n < 2) return fail(c, "gyro: no rates uploaded");
    bool bad = false;
    for (double v : c->g_ts) bad = bad || !std::isfinite(v);
    for (double v : c->g_rates) bad = bad || !std::isfinite(v);
    std::vector<double> q(4 * (size_t)n);
    std::vector<int64_t> us(n);
    double cur[4] = {1., 0., 0., 0.};
    for (uint32_t i = 0; i < n; ++i) {
        us[i] = std::isfinite(c->g_ts[i]) ? (int64_t)(c->g_ts[i] * 1000000) : 0;
        if (i > 0) {
            const double dt = c->g_ts[i] - c->g_ts[i - 1];
            const double* r = &c->g_rates[3 * (size_t)i];
            const double w[3] = {r[axis[0]] * sign[0] * dt, r[axis[1]] * sign[1] * dt, r[axis[2]] * sign[2] * dt};
            double d[4];
            rs::gyro_delta(w, d);
            rs::quat_mul_norm(d, cur);
        }
        for (int k = 0; k < 4; ++k) q[4 * (size_t)i + k] = cur[k];
    }
    return timestamped(c, us.data(), q.data(), n, bad, out);
}

// (the stand-in computes at once; the status is kept for rship_gyro_batch_status as the device keeps its records)
int rship_gyro_rates_integrate_enqueue(rship_ctx* c, const int32_t axis[3], const double sign[3], uint32_t slot, rship_gyro_result* out) {
    if (!slot || slot >= 256) return fail(c, "gyro: status slot out of range");
    const int rc = rship_gyro_rates_integrate(c, axis, sign, out);
    if (rc) return rc;
    if (c->batch_gyro_status.size() < slot) c->batch_gyro_status.resize(slot, RSHIP_GYRO_OK);
    c->batch_gyro_status[slot - 1] = out->status;
    return 0;
}
int rship_gyro_batch_status(rship_ctx* c, uint32_t n, int32_t* status) {
    for (uint32_t i = 0; i < n; ++i) status[i] = i < c->batch_gyro_status.size() ? c->batch_gyro_status[i] : RSHIP_GYRO_OK;
    return 0;
}

int rship_gyro_knots(rship_ctx* c, double* out, uint32_t cap_knots) {
    if ((size_t)cap_knots * 4 < c->knots.size()) return fail(c, "gyro_knots: buffer too small");
    std::copy(c->knots.begin(), c->knots.end(), out);
    return 0;
}

int rship_gyro_table(rship_ctx* c, double* out16, uint32_t cap_knots) {
    if ((size_t)cap_knots * 16 < c->coef64.size()) return fail(c, "gyro_table: buffer too small");
    std::copy(c->coef64.begin(), c->coef64.end(), out16);
    return 0;
}

void* rship_host_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
void rship_host_free(void* p) { std::free(p); }

int rship_upload_raw(rship_ctx* c, const double* host, uint64_t arena_offset, uint64_t n_doubles) {
    if (c->raw.size() < arena_offset + n_doubles) c->raw.resize(arena_offset + n_doubles);
    std::memcpy(c->raw.data() + arena_offset, host, (size_t)n_doubles * 8);
    return 0;
}

// pack_frames_kernel, one pair at a time (same headers, same operations)
int rship_pack_frames(rship_ctx* c, const rship_frame* table, const rship_pack_frame* pack, uint32_t nf, uint64_t total,
                      double start, double fs, uint32_t* bad) {
    c->rays_a.assign(total, f4{0, 0, 0, 0});
    c->rays_b.assign(total, f4{0, 0, 0, 0});
    for (auto& q : c->q) q.assign(2 * total, 0.0);
    c->frames.assign(table, table + nf);
    c->sel.clear();
    c->grp.clear();
    c->grp_off.assign(2, 0);
    uint32_t nb = 0;
    for (uint32_t fi = 0; fi < nf; ++fi) {
        const rship_pack_frame& fr = pack[fi];
        const uint64_t rec_len = (uint64_t)fr.n_rays * (fr.is_pixels ? 4 : 8);
        if (fr.raw_offset + rec_len > c->raw.size() || (uint64_t)fr.ray_offset + fr.n_rays > total)
            return fail(c, "pack: record outside a buffer");
        const double* rec = c->raw.data() + fr.raw_offset;
        const uint32_t n = fr.n_rays;
        const rs::Lens lens{fr.lens[0], fr.lens[1], fr.lens[2], fr.lens[3], fr.lens[4], fr.lens[5], fr.lens[6], fr.lens[7], fr.lens[8]};
        for (uint32_t i = 0; i < n; ++i) {
            double ra[3], rb[3], tsa, tsb;
            if (fr.is_pixels) {
                const double* q = rec + 4 * (size_t)i;
                rs::pixel_to_ray(lens, q[0], q[1], fr.time_a, fr.rows, ra, &tsa);
                rs::pixel_to_ray(lens, q[2], q[3], fr.time_b, fr.rows, rb, &tsb);
            } else {
                tsa = rec[i];
                tsb = rec[(size_t)n + i];
                for (int k = 0; k < 3; ++k) { ra[k] = rec[2 * (size_t)n + 3 * (size_t)i + k]; rb[k] = rec[5 * (size_t)n + 3 * (size_t)i + k]; }
            }
            const double ta = rs::knot_offset(tsa, start, fs, fr.base), tb = rs::knot_offset(tsb, start, fs, fr.base);
            const f4 o0{(float)ra[0], (float)rb[0], (float)ra[1], (float)rb[1]};
            const f4 o1{(float)ra[2], (float)rb[2], (float)ta, (float)tb};
            const float v[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
            bool ok = true;
            for (float x : v) ok = ok && std::isfinite(x);
            nb += ok ? 0u : 1u;
            const size_t o = (size_t)fr.ray_offset + i;
            c->rays_a[o] = o0;
            c->rays_b[o] = o1;
            for (int k = 0; k < 3; ++k) { c->q[k][2 * o] = ra[k]; c->q[k][2 * o + 1] = rb[k]; }
            c->q[3][2 * o] = ta;
            c->q[3][2 * o + 1] = tb;
        }
    }
    if (bad) *bad = nb;
    return 0;
}

int rship_set_problem_frames(rship_ctx*, const rship_frame*, uint32_t) { return 0; } // (window planning: nothing to plan here)

// the native exchange needs a GPU and librccl: not part of the test double
int rship_rccl_preflight(rship_ctx* c) { return fail(c, "rccl: device only"); }
const char* rship_rccl_library(rship_ctx*) { return ""; }
int rship_rccl_unique_id(rship_ctx* c, void*) { return fail(c, "rccl: device only"); }
int rship_rccl_init(rship_ctx* c, const void*, int, int) { return fail(c, "rccl: device only"); }
int rship_rccl_allreduce(rship_ctx* c, double*, uint64_t) { return fail(c, "rccl: device only"); }
int rship_rccl_shutdown(rship_ctx* c) { return fail(c, "rccl: device only"); }
uint64_t rship_loop_exchanges(const rship_ctx*) { return 0; }
int rship_set_loop_exchange(rship_ctx*, rship_loop_exchange_fn, void*) { return 0; }

int rship_debug_rays(rship_ctx* c, uint32_t frame_index, float* a4, float* b4, uint32_t cap) {
    if (frame_index >= c->frames.size()) return fail(c, "debug_rays: index out of range");
    const rship_frame& fr = c->frames[frame_index];
    if (fr.n_rays > cap) return fail(c, "debug_rays: output too small");
    std::memcpy(a4, c->rays_a.data() + fr.ray_offset, (size_t)fr.n_rays * 16);
    std::memcpy(b4, c->rays_b.data() + fr.ray_offset, (size_t)fr.n_rays * 16);
    return 0;
}

int rship_select_slots(rship_ctx* c, const uint32_t* idx, uint32_t n, const uint32_t* grp_off, uint32_t n_grp) {
    if (n_grp < 1) n_grp = 1;
    c->sel.assign(idx, idx + n);
    c->grp.assign(n, 0);
    c->grp_off.assign(n_grp + 1, 0);
    if (grp_off) {
        c->grp_off.assign(grp_off, grp_off + n_grp + 1);
        for (uint32_t w = 0; w < n_grp; ++w)
            for (uint32_t j = grp_off[w]; j < grp_off[w + 1]; ++j) c->grp[j] = w;
    } else {
        c->grp_off[n_grp] = n;
    }
    c->M.assign((size_t)n * 3, 0.0);
    c->k.assign(n, 0.0);
    c->init_h.assign(n, kNone);
    return 0;
}
int rship_select_frames(rship_ctx* c, const uint32_t* idx, uint32_t n) { return rship_select_slots(c, idx, n, nullptr, 1); }

int rship_set_plan(rship_ctx* c, const uint32_t* plan_idx, uint32_t plan_len, const uint32_t* chunk_off, uint32_t n_chunks,
                   const uint32_t* win_chunk_off, uint32_t n_win) {
    if (n_win < 1 || win_chunk_off[0] != 0 || win_chunk_off[n_win] != n_chunks) return fail(c, "plan: bad window offsets");
    if (n_chunks && (chunk_off[0] != 0 || chunk_off[n_chunks] != plan_len)) return fail(c, "plan: bad chunk offsets");
    c->plan_has_idx = plan_idx != nullptr;
    c->plan_idx.assign(plan_idx ? plan_idx : chunk_off, plan_idx ? plan_idx + plan_len : chunk_off);
    if (n_chunks) c->plan_chunk_off.assign(chunk_off, chunk_off + n_chunks + 1);
    else c->plan_chunk_off.assign(1, 0u);
    c->plan_win_off.assign(win_chunk_off, win_chunk_off + n_win + 1);
    return 0;
}

namespace {
// plan_sum_kernel: chunks summed sequentially, windows = sequential sum of their chunks
void plan_sum(rship_ctx* c, const std::vector<double>& in, uint32_t rows, size_t cols) {
    const size_t nc = c->plan_chunk_off.size() - 1, nw = c->plan_win_off.size() - 1;
    c->chunk_out.assign((size_t)rows * nc + 1, 0.0);
    c->win_out.assign((size_t)rows * nw + 1, 0.0);
    for (uint32_t r = 0; r < rows; ++r)
        for (size_t w = 0; w < nw; ++w) {
            double tot = 0.0;
            for (uint32_t ch = c->plan_win_off[w]; ch < c->plan_win_off[w + 1]; ++ch) {
                double acc = 0.0;
                for (uint32_t j = c->plan_chunk_off[ch]; j < c->plan_chunk_off[ch + 1]; ++j)
                    acc += in[(size_t)r * cols + (c->plan_has_idx ? c->plan_idx[j] : j)];
                c->chunk_out[(size_t)r * nc + ch] = acc;
                tot += acc;
            }
            c->win_out[(size_t)r * nw + w] = tot;
        }
}
} // namespace

int rship_presync_enqueue(rship_ctx* c, const int32_t* kd, const float* fd, const int32_t* /*kd64*/, const double* /*fd64*/, uint32_t n_cand, uint32_t n_hyp,
                          uint32_t stream_base, uint64_t seed, int, int) {
    uint32_t fl = 0;
    c->pend_rows = 0;
    const size_t ns = c->sel.size();
    if (!n_cand || !ns) return 0;
    c->frame_cost.assign((size_t)n_cand * ns, 0.0);
    c->best_h.assign((size_t)n_cand * ns, 0);
    for (uint32_t ci = 0; ci < n_cand; ++ci) {
        for (size_t s = 0; s < ns; ++s) {
            const rship_frame& fr = c->frames[c->sel[s]];
            Rows t = unit_rows(c, fr, kd[ci], fd[ci]);
            if (t.bad) fl |= RSHIP_BAD_P;
            f3 Mv;
            int bh = lmeds(t, n_hyp, seed, fr.id, stream_base + ci, Mv);
            if (!(std::isfinite(Mv.x) && std::isfinite(Mv.y) && std::isfinite(Mv.z))) fl |= RSHIP_BAD_M;
            double ss = 0;
            std::vector<float> pm(fr.n_rays);
            for (uint32_t i = 0; i < fr.n_rays; ++i) {
                pm[i] = t.nrm[i] * rs::dot(t.n[i], Mv);
                ss += (double)pm[i] * pm[i];
            }
            float kf = clampk(100.0f / std::sqrt((float)ss));
            float sc = kf / std::sqrt(rs::dot(Mv, Mv));
            double acc = 0;
            for (uint32_t i = 0; i < fr.n_rays; ++i) {
                float r = pm[i] * sc;
                if (!std::isfinite(r)) fl |= RSHIP_BAD_R;
                float rho = rs::log1p_pos(r * r);
                if (!std::isfinite(rho)) fl |= RSHIP_BAD_RHO;
                acc += std::sqrt(rho);
            }
            c->frame_cost[(size_t)ci * ns + s] = std::sqrt(acc);
            c->best_h[(size_t)ci * ns + s] = bh;
        }
    }
    plan_sum(c, c->frame_cost, n_cand, ns);
    c->pend_rows = n_cand;
    c->pend_flags = fl;
    if (c->batch_n) { // keep this sweep's sums for rship_presync_batch_collect
        c->batch_win.push_back(c->win_out);
        c->batch_chunk.push_back(c->chunk_out);
        c->batch_flags.push_back(fl);
    }
    return 0;
}

int rship_presync_batch_begin(rship_ctx* c, uint32_t n, uint32_t n_cand) { // (n = 0: cancel)
    c->batch_n = n;
    c->batch_rows = n_cand;
    c->batch_win.clear(); c->batch_chunk.clear(); c->batch_flags.clear();
    c->pend_rows = 0;
    return 0;
}
int rship_presync_batch_collect(rship_ctx* c, uint32_t n_cand, double* win_costs, double* chunk_costs, uint32_t* flags) {
    const size_t nc = c->plan_chunk_off.size() - 1, nw = c->plan_win_off.size() - 1;
    const uint32_t n = c->batch_n;
    c->batch_n = 0;
    if (!n) { c->err = "presync batch: none open"; return 1; }
    if (flags) std::fill(flags, flags + n, 0u);
    if (win_costs) std::fill(win_costs, win_costs + (size_t)n * n_cand * nw, 0.0);
    if (chunk_costs) std::fill(chunk_costs, chunk_costs + (size_t)n * n_cand * nc, 0.0);
    if (c->batch_win.empty()) return 0; // (no candidates or no slots on this device)
    if (c->batch_win.size() != n) { c->err = "presync batch: fewer sweeps were enqueued than the batch was opened for"; return 1; }
    for (uint32_t b = 0; b < n; ++b) {
        if (win_costs) std::copy(c->batch_win[b].begin(), c->batch_win[b].begin() + (size_t)n_cand * nw, win_costs + (size_t)b * n_cand * nw);
        if (chunk_costs) std::copy(c->batch_chunk[b].begin(), c->batch_chunk[b].begin() + (size_t)n_cand * nc, chunk_costs + (size_t)b * n_cand * nc);
        if (flags) flags[b] = c->batch_flags[b];
    }
    c->pend_rows = 0;
    return 0;
}

int rship_presync_collect(rship_ctx* c, uint32_t n_cand, double* win_costs, double* chunk_costs, uint32_t* flags,
                          double* frame_costs, int32_t* best_h) {
    const size_t nc = c->plan_chunk_off.size() - 1, nw = c->plan_win_off.size() - 1, ns = c->sel.size();
    if (flags) *flags = 0;
    if (!c->pend_rows) {
        if (win_costs) std::fill(win_costs, win_costs + (size_t)n_cand * nw, 0.0);
        if (chunk_costs) std::fill(chunk_costs, chunk_costs + (size_t)n_cand * nc, 0.0);
        return 0;
    }
    if (win_costs) std::copy(c->win_out.begin(), c->win_out.begin() + (size_t)n_cand * nw, win_costs);
    if (chunk_costs) std::copy(c->chunk_out.begin(), c->chunk_out.begin() + (size_t)n_cand * nc, chunk_costs);
    if (flags) *flags = c->pend_flags;
    if (frame_costs) std::copy(c->frame_cost.begin(), c->frame_cost.begin() + (size_t)n_cand * ns, frame_costs);
    if (best_h) std::copy(c->best_h.begin(), c->best_h.begin() + (size_t)n_cand * ns, best_h);
    c->pend_rows = 0;
    return 0;
}

int rship_init_motion(rship_ctx* c, const int32_t* kd, const float* fd, const int32_t* /*kd64*/, const double* /*fd64*/, uint32_t n_hyp, uint32_t stream,
                      uint32_t stream_stride, uint64_t seed) {
    c->init_h.assign(c->sel.size(), kNone);
    for (size_t s = 0; s < c->sel.size(); ++s) {
        const uint32_t fi = c->sel[s], g = c->grp[s];
        const rship_frame& fr = c->frames[fi];
        Rows t = unit_rows(c, fr, kd[g], fd[g]);
        f3 Mv;
        c->init_h[s] = lmeds(t, n_hyp, seed, fr.id, stream + g * stream_stride, Mv);
    }
    c->init_seed = seed;
    c->init_stream = stream;
    c->init_stride = stream_stride;
    return 0;
}

namespace {

// wave_sum_f64 of kernels/common.hpp on 64 lane values: four row_shr steps inside the rows of 16 (a lane whose
// source is outside its row adds 0), then the four row totals as (R3 + R2) + (R1 + R0)
double wave_sum_order(const double* lanes) {
    double v[64], t[64];
    for (int l = 0; l < 64; ++l) v[l] = lanes[l];
    for (int sh = 1; sh <= 8; sh *= 2) {
        for (int l = 0; l < 64; ++l) t[l] = v[l] + ((l & 15) >= sh ? v[l - sh] : 0.0);
        for (int l = 0; l < 64; ++l) v[l] = t[l];
    }
    return (v[63] + v[47]) + (v[31] + v[15]);
}

// the workgroup shape launch_motion64 picks (rows per thread, waves) for a frame of n tracks: its size class, whatever
// else the problem holds (a thread adds its rows in order and rows beyond the frame add zeros, so inside the one-wave /
// four-wave family the sums do not depend on the rows per thread)
void motion_shape(uint32_t n, int& rpt, int& nw) {
    if (n <= 64) { rpt = 1; nw = 1; }
    else if (n <= 128) { rpt = 2; nw = 1; }
    else if (n <= 192) { rpt = 3; nw = 1; }
    else if (n <= 256) { rpt = 4; nw = 1; }
    else if (n <= 512) { rpt = 8; nw = 1; }
    else { nw = 4; rpt = 4; while ((uint32_t)rpt * 256u < n) rpt *= 2; }
}

// sum over a workgroup of per-thread values in the kernels' order: wave sums, then the waves left to right
double block_sum_order(const std::vector<double>& per_thread, int nw) {
    double tot = 0.0;
    for (int w = 0; w < nw; ++w) {
        const double sw = wave_sum_order(per_thread.data() + 64 * w);
        tot = w == 0 ? sw : tot + sw;
    }
    return tot;
}

// opt_motion64_kernel's prologue: the fp64 rows as the workgroup holds them (row j * threads + tid in thread
// tid's register j; zero beyond the frame), GuessMotion's winner -> M in fp64, GuessK
void finish_slot(rship_ctx* c, size_t sl, int32_t kd, double fd, bool simple_k, int rpt, int nw, std::vector<d3>& P) {
    const uint32_t fi = c->sel[sl], grp = c->grp[sl];
    const rship_frame& fr = c->frames[fi];
    const int threads = 64 * nw;
    P.assign((size_t)rpt * threads, d3{0, 0, 0});
    for (uint32_t i = 0; i < fr.n_rays; ++i) row64(c, fr, i, kd, fd, P[i], nullptr);
    const int32_t pend = sl < c->init_h.size() ? c->init_h[sl] : kNone;
    if (!simple_k && pend == kNone) return;
    d3 Mv{0, 0, 0};
    if (!simple_k && pend >= 0) {
        uint32_t i0, i1;
        rs::sample_pair(c->init_seed, fr.id, c->init_stream + grp * c->init_stride, (uint32_t)pend, fr.n_rays, i0, i1);
        Mv = rs::cross(P[i0], P[i1]);
        const double nn = std::sqrt(rs::dot(Mv, Mv));
        if (!(nn < 1e-12)) Mv = rs::scale(Mv, 1.0 / nn);
    }
    std::vector<double> part((size_t)threads, 0.0);
    for (int t = 0; t < threads; ++t) {
        double ss = 0.0;
        for (int j = 0; j < rpt; ++j) {
            const d3& p = P[(size_t)j * threads + t];
            const double pm = simple_k ? std::sqrt(rs::dot(p, p)) : rs::dot(p, Mv);
            ss = std::fma(pm, pm, ss);
        }
        part[t] = ss;
    }
    const double tot = block_sum_order(part, nw);
    if (!simple_k) { c->M[3 * sl] = Mv.x; c->M[3 * sl + 1] = Mv.y; c->M[3 * sl + 2] = Mv.z; }
    c->k[sl] = rs::clamp_k64(100.0 / std::sqrt(tot));
    if (sl < c->init_h.size()) c->init_h[sl] = kNone;
}

// MotionEval64 of kernels/sync64.hpp: the row terms of rs::motion_row per thread, the four sums in the
// workgroup's order
struct MotionEvalCpu {
    const std::vector<d3>* P;
    int rpt, nw;
    double k2;
    int evals = 0;
    double operator()(const double x[3], double g[3]) {
        const int threads = 64 * nw;
        double inv_xx;
        const double inv_s = rs::motion_inv_s(x, k2, &inv_xx);
        std::vector<double> part[4];
        for (auto& v : part) v.assign((size_t)threads, 0.0);
        for (int t = 0; t < threads; ++t) {
            double L = 0.0, a0 = 0.0, a1 = 0.0, a2 = 0.0;
            for (int j = 0; j < rpt; ++j) rs::motion_row((*P)[(size_t)j * threads + t], x, inv_s, L, a0, a1, a2);
            part[0][t] = L; part[1][t] = a0; part[2][t] = a1; part[3][t] = a2;
        }
        double tsum[4];
        for (int q = 0; q < 4; ++q) tsum[q] = block_sum_order(part[q], nw);
        ++evals;
        return rs::motion_finish(x, inv_xx, tsum, g);
    }
};

struct LbfgsHistCpu {
    double s_S[rs::kLbfgsBasis][3], s_Y[rs::kLbfgsBasis][3], s_inv_ys[rs::kLbfgsBasis], s_rho[rs::kLbfgsBasis], s_alpha[rs::kLbfgsBasis];
    const double* S(int i) const { return s_S[i]; }
    const double* Y(int i) const { return s_Y[i]; }
    double inv_ys(int i) const { return s_inv_ys[i]; }
    double& rho(int i) { return s_rho[i]; }
    double& alpha(int i) { return s_alpha[i]; }
    void store(int op, const double sv[3], const double yv[3]) {
        for (int c = 0; c < 3; ++c) { s_S[op][c] = sv[c]; s_Y[op][c] = yv[c]; }
        s_inv_ys[op] = 1.0 / rs::dot3(yv, sv);
    }
};

// opt_motion64_kernel, one slot after the other
void motion_pass(rship_ctx* c, const int32_t* kdv, const double* fdv, int max_iters, bool simple_k, uint64_t* stats) {
    uint64_t tot_it = 0, tot_ev = 0, tot_bnl = 0;
    for (size_t sl = 0; sl < c->sel.size(); ++sl) {
        int rpt = 1, nw = 1;
        motion_shape(c->frames[c->sel[sl]].n_rays, rpt, nw);
        const uint32_t grp = c->grp[sl];
        const int32_t kd = kdv[grp];
        const double fd = fdv[grp];
        if (fd != fd) continue; // window switched off
        std::vector<d3> P;
        finish_slot(c, sl, kd, fd, simple_k, rpt, nw, P);
        if (max_iters <= 0 || simple_k) continue;
        MotionEvalCpu ev{&P, rpt, nw, c->k[sl] * c->k[sl]};
        LbfgsHistCpu hist;
        double x[3] = {c->M[3 * sl], c->M[3 * sl + 1], c->M[3 * sl + 2]};
        int bnl = 0;
        const int it = rs::lbfgs3(ev, hist, x, max_iters, c->lbfgs_reeval, &bnl);
        c->M[3 * sl] = x[0]; c->M[3 * sl + 1] = x[1]; c->M[3 * sl + 2] = x[2];
        tot_it += (uint64_t)it;
        tot_ev += (uint64_t)ev.evals;
        tot_bnl += (uint64_t)bnl;
    }
    if (stats) { stats[0] = tot_it; stats[1] = tot_ev; stats[2] = tot_bnl; }
}
} // namespace

int rship_opt_motion(rship_ctx* c, const int32_t* kd, const double* fd, uint64_t* stats) {
    motion_pass(c, kd, fd, 200, false, stats);
    return 0;
}
int rship_finish_init(rship_ctx* c, const int32_t* kd, const double* fd) {
    motion_pass(c, kd, fd, 0, false, nullptr);
    return 0;
}
int rship_init_k_simple(rship_ctx* c, const int32_t* kd, const double* fd) {
    motion_pass(c, kd, fd, 0, true, nullptr);
    return 0;
}

int rship_opt_motion_detail(rship_ctx* c, const int32_t*, const double*, uint32_t*, uint32_t) { return fail(c, "opt_motion_detail: device only"); }

int rship_loss_enqueue(rship_ctx* c, const int32_t* kd, const double* fd, uint32_t n_delays, int want_grad, uint32_t flags) {
    const bool simple = (flags & RSHIP_LOSS_SIMPLIFIED) != 0;
    const size_t ng = c->grp_off.size() - 1, ns = c->sel.size();
    c->pend_rows = 0;
    if (!n_delays || !ns) return 0;
    const uint32_t rows = want_grad ? 2 * n_delays : n_delays;
    std::vector<double> part((size_t)rows * ns, 0.0); // [loss rows][grad rows] x slots, like the kernel
    for (uint32_t b = 0; b < n_delays; ++b) {
        for (size_t w = 0; w < ng; ++w) {
            const int32_t kdw = kd[b * ng + w];
            const double fdw = fd[b * ng + w];
            for (uint32_t sl = c->grp_off[w]; fdw == fdw && sl < c->grp_off[w + 1]; ++sl) {
                const rship_frame& fr = c->frames[c->sel[sl]];
                const double Mx = simple ? 0.0 : c->M[3 * sl], My = simple ? 0.0 : c->M[3 * sl + 1], Mz = simple ? 0.0 : c->M[3 * sl + 2];
                const double kk = c->k[sl];
                const d3 Mv{Mx, My, Mz};
                const double inv_s = rs::loss_inv_s(simple, kk, Mv);
                // loss64_kernel: 256 threads, thread tid takes rows tid, tid + 256, ...; wave sums; waves in order
                std::vector<double> Lt(256, 0.0), Gt(256, 0.0);
                for (uint32_t i = 0; i < fr.n_rays; ++i) {
                    d3 P, dP{0, 0, 0};
                    row64(c, fr, i, kdw, fdw, P, want_grad ? &dP : nullptr);
                    double& L = Lt[i & 255u];
                    double& G = Gt[i & 255u];
                    if (simple) { if (want_grad) rs::loss_row<true, true>(P, dP, Mv, inv_s, L, G); else rs::loss_row<false, true>(P, dP, Mv, inv_s, L, G); }
                    else { if (want_grad) rs::loss_row<true, false>(P, dP, Mv, inv_s, L, G); else rs::loss_row<false, false>(P, dP, Mv, inv_s, L, G); }
                }
                part[(size_t)b * ns + sl] = block_sum_order(Lt, 4);
                if (want_grad) part[(size_t)(n_delays + b) * ns + sl] = block_sum_order(Gt, 4) * c->fs;
            }
        }
    }
    plan_sum(c, part, rows, ns);
    c->pend_rows = rows;
    c->pend_grad = want_grad != 0;
    return 0;
}

int rship_loss_collect(rship_ctx* c, uint32_t n_delays, double* win_loss, double* win_grad, double* chunk_loss,
                       double* chunk_grad) {
    const size_t nc = c->plan_chunk_off.size() - 1, nw = c->plan_win_off.size() - 1;
    const size_t wn = (size_t)n_delays * nw, cn = (size_t)n_delays * nc;
    if (!c->pend_rows) {
        if (win_loss) std::fill(win_loss, win_loss + wn, 0.0);
        if (win_grad) std::fill(win_grad, win_grad + wn, 0.0);
        if (chunk_loss) std::fill(chunk_loss, chunk_loss + cn, 0.0);
        if (chunk_grad) std::fill(chunk_grad, chunk_grad + cn, 0.0);
        return 0;
    }
    if (win_loss) std::copy(c->win_out.begin(), c->win_out.begin() + wn, win_loss);
    if (win_grad && c->pend_grad) std::copy(c->win_out.begin() + wn, c->win_out.begin() + 2 * wn, win_grad);
    if (chunk_loss) std::copy(c->chunk_out.begin(), c->chunk_out.begin() + cn, chunk_loss);
    if (chunk_grad && c->pend_grad) std::copy(c->chunk_out.begin() + cn, c->chunk_out.begin() + 2 * cn, chunk_grad);
    c->pend_rows = 0;
    return 0;
}

int rship_exec_supported(rship_ctx*) { return 0; } // the window executor is a device scheduler: nothing to stand in for
int rship_exec_stats(rship_ctx*, uint32_t out[4]) { out[0] = out[1] = out[2] = out[3] = 0; return 0; }
int rship_debug_residuals(rship_ctx* c, int, uint32_t) { c->err = "debug_residuals: not in the CPU stand-in"; return 1; }
int rship_debug_residuals_get(rship_ctx* c, uint32_t*, uint64_t, uint32_t dims[4]) { dims[0] = dims[1] = dims[2] = dims[3] = 0; c->err = "debug_residuals: not in the CPU stand-in"; return 1; }
int rship_lmeds_shapes(rship_ctx*, uint32_t out[6]) { for (int k = 0; k < 6; ++k) out[k] = 0; return 0; } // (no tile kernel here)
int rship_near_static_stats(rship_ctx*, uint64_t out[3]) { out[0] = out[1] = out[2] = 0; return 0; } // (the stand-in's sweep is its own sequential fp32 search)
int rship_window_info(rship_ctx*, uint32_t out[8]) { for (int i = 0; i < 8; ++i) out[i] = 0; return 0; }
int rship_sync_exec(rship_ctx* c, const double*, int, uint32_t, uint32_t, uint64_t, int, double, double, double*, double*, int32_t*,
                    double*, uint32_t) {
    return fail(c, "sync_exec: device only");
}
int rship_has_device_loop(void) { return 0; } // the host loop is what this double is there to exercise
int rship_sync_run(rship_ctx* c, const double*, int, double, double, int, double*, int32_t*, double*) {
    return fail(c, "sync_run: device only");
}

int rship_get_motion(rship_ctx* c, double* M, double* k, uint32_t cap, uint32_t* n) {
    uint32_t cnt = (uint32_t)std::min<size_t>(c->sel.size(), cap);
    for (uint32_t i = 0; i < cnt; ++i) {
        M[3 * i] = c->M[3 * i]; M[3 * i + 1] = c->M[3 * i + 1]; M[3 * i + 2] = c->M[3 * i + 2];
        k[i] = c->k[i];
    }
    if (n) *n = cnt;
    return 0;
}

int rship_set_motion(rship_ctx* c, const double* M, const double* k, uint32_t n) {
    if (n != c->sel.size()) return fail(c, "set_motion: count differs from the selection");
    c->M.assign(M, M + 3 * (size_t)n);
    c->k.assign(k, k + n);
    return 0;
}

int rship_debug_problem(rship_ctx* c, uint32_t sel_index, int32_t kd, float fd, float* P, float* dP, uint32_t cap_rows) {
    if (sel_index >= c->sel.size()) return fail(c, "debug_problem: index out of range");
    const rship_frame& fr = c->frames[c->sel[sel_index]];
    if (fr.n_rays > cap_rows) return fail(c, "debug_problem: output too small");
    for (uint32_t i = 0; i < fr.n_rays; ++i) {
        f3 p, d;
        row(c, fr, i, kd, fd, p, dP ? &d : nullptr);
        P[3 * i] = p.x; P[3 * i + 1] = p.y; P[3 * i + 2] = p.z;
        if (dP) { dP[3 * i] = d.x * (float)c->fs; dP[3 * i + 1] = d.y * (float)c->fs; dP[3 * i + 2] = d.z * (float)c->fs; }
    }
    return 0;
}

int rship_debug_problem64(rship_ctx* c, uint32_t sel_index, int32_t kd, double fd, double* P, double* dP, uint32_t cap_rows) {
    if (sel_index >= c->sel.size()) return fail(c, "debug_problem: index out of range");
    const rship_frame& fr = c->frames[c->sel[sel_index]];
    if (fr.n_rays > cap_rows) return fail(c, "debug_problem: output too small");
    for (uint32_t i = 0; i < fr.n_rays; ++i) {
        d3 p, d;
        row64(c, fr, i, kd, fd, p, dP ? &d : nullptr);
        P[3 * i] = p.x; P[3 * i + 1] = p.y; P[3 * i + 2] = p.z;
        if (dP) { dP[3 * i] = d.x * c->fs; dP[3 * i + 1] = d.y * c->fs; dP[3 * i + 2] = d.z * c->fs; }
    }
    return 0;
}

int rship_debug_math64(rship_ctx* c, int op, const double* a, const double* b, double* out, uint32_t n) {
    if (op < 0 || op > 5 || !n) return fail(c, "debug_math64: bad arguments");
    if (op == 4) {
        for (uint32_t blk = 0; blk < (n + 63) / 64; ++blk) {
            double lanes[64];
            for (uint32_t l = 0; l < 64; ++l) lanes[l] = blk * 64 + l < n ? a[blk * 64 + l] : 0.0;
            out[blk] = wave_sum_order(lanes);
        }
        return 0;
    }
    for (uint32_t i = 0; i < n; ++i) {
        const double x = a[i], y = b ? b[i] : 0.0;
        if (op == 0) out[i] = x / y;
        else if (op == 1) out[i] = std::sqrt(x);
        else if (op == 2) { double rc; out[2 * i] = rs::log1p_rcp_f64(x, &rc); out[2 * i + 1] = rc; }
        else if (op == 5) out[i] = rs::div3_exact(x);
        else out[i] = std::fma(x, y, x);
    }
    return 0;
}

int rship_debug_init_h(rship_ctx* c, int32_t* get, const int32_t* set, uint32_t n) {
    if (n != c->sel.size() || c->init_h.size() != n) return fail(c, "debug_init_h: count differs from the selection");
    if (get) std::copy(c->init_h.begin(), c->init_h.end(), get);
    if (set) c->init_h.assign(set, set + n);
    return 0;
}

int rship_debug_k2_counters(rship_ctx*, uint64_t out[16], int) {
    for (int i = 0; i < 16; ++i) out[i] = 0;
    return 0;
}
int rship_debug_select(rship_ctx* c, const float*, uint32_t, uint32_t, uint32_t, const float*, uint32_t*) {
    return fail(c, "debug_select: device only");
}
int rship_profile_enable(rship_ctx*, int) { return 0; }
int rship_profile_get(rship_ctx*, int, uint64_t* launches, double* total_ms) {
    if (launches) *launches = 0;
    if (total_ms) *total_ms = 0;
    return 0;
}
int rship_profile_reset(rship_ctx*) { return 0; }

} // extern "C"
