// handoff_probe.hip -- does the window executor's cross-workgroup hand-off protocol (kernels/executor.hpp) deliver
// stale words at EIGHT ONE-WAVE WORKGROUPS PER CU?  MI355X_MICROARCH.md ("inter-workgroup visibility") lists the form
// -- sc1 stores -> s_waitcnt vmcnt(0) -> agent-scope atomic add / sc1 cell store; the consumer's sc1 poll or the
// value its own add returned -> sc1 loads -- as MEASURED at one workgroup per CU, not as an architectural guarantee.
// This is the same measurement in the executor's own geometry, "under uneven load, checking every word" as the guide
// asks: a miniature of the executor with nothing but tagged data.
//
//   G groups (windows) x S slots (frames); generation n of a group = S tasks, one per slot, pulled from a ring of
//   lap-tagged 64-bit cells by persistent one-wave workgroups (exec_pop / exec_push, copied in form).  A task
//     1. loads the group's record (16 x 8 B, written by the wave that pushed the generation) and checks every word;
//     2. loads its slot's payload of the previous generation (16 x 8 B, written by whichever wave ran that task) and
//        checks every word;
//     3. sleeps a pseudo-random time (uneven load), stores the payload of this generation (sc1), waits for its stores,
//        subtracts one from the group's counter (agent-scope atomic);
//     4. the wave whose subtraction came last loads ALL S payloads of the group and checks every word, stores the
//        next generation's record, reserves queue numbers, waits for its stores, stores the cells.
//   Every expected value is a function of (group or slot, generation, word), so one stale word is one count.
//
// Second question (executor.hpp: "an sc1 load of the 'windows done' word was seen to return a stale value for
// seconds"): idle waves read a monotonic counter that only atomics change first with an atomic, then with an sc1
// load; a load that returns LESS than the earlier atomic read is a stale read.  Counted for the counter on a line
// of its own and on the line it shares with the queue's head and tail (the layout in which the observation was made).
//
// Every wait has a time limit (s_memrealtime) after which the wave raises the abort flag and everybody leaves.
//
//   hipcc --offload-arch=gfx950 -O2 -o handoff_probe tools/ubench/handoff_probe.hip && ./handoff_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(2); } } while (0)

namespace {

constexpr int kWords = 16;
constexpr uint32_t kSlotBits = 13; // a cell's low word: generation << 13 | slot

struct Probe {
    unsigned long long* q;
    uint32_t q_mask, q_shift;
    uint32_t* q_head; uint32_t* q_tail; uint32_t* done; uint32_t* abort_flag;
    unsigned long long* grec;     // [G][kWords]: word 1 = the group's counter (low 32 bits), the others tagged
    unsigned long long* payload;  // [G * S][kWords]
    uint32_t G, S, gens;
    unsigned long long* stats;    // see main()
    unsigned long long watchdog_ticks;
    uint32_t sleep_mask;          // task length: (hash & sleep_mask) x s_sleep 8
    uint32_t poll_mask;           // idle waves look at the counters every poll_mask + 1 reads of their cell
};

template <class T> __device__ __forceinline__ T ld_sc1(const T* p) { return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ void st_sc1(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ unsigned long long tag(uint32_t kind, uint32_t id, uint32_t gen, uint32_t word) {
    unsigned long long z = ((unsigned long long)kind << 60) ^ ((unsigned long long)id << 36) ^ ((unsigned long long)gen << 8) ^ word;
    z = (z ^ (z >> 31)) * 0x9E3779B97F4A7C15ull; // (mixed, so that a torn or shifted word never matches by accident)
    return z ^ (z >> 29);
}

__device__ __forceinline__ void count(const Probe& p, int i, unsigned long long n) {
    if (n) atomicAdd(&p.stats[i], n);
}

__device__ uint32_t pop(const Probe& p) {
    uint32_t idx = 0;
    if (threadIdx.x == 0) idx = __hip_atomic_fetch_add(p.q_head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    idx = uni(idx);
    const unsigned long long* cell = p.q + (idx & p.q_mask);
    const uint32_t lap = (idx >> p.q_shift) + 1u;
    uint32_t seen_tail = 0xffffffffu;
    unsigned long long t_push = __builtin_amdgcn_s_memrealtime();
    for (uint32_t spins = 0;; ++spins) {
        const unsigned long long v = ld_sc1(cell);
        if (uni((uint32_t)(v >> 32)) == lap) return uni((uint32_t)v);
        if ((spins & p.poll_mask) == p.poll_mask) {
            uint32_t v3 = 0;
            if (threadIdx.x < 3)
                v3 = __hip_atomic_fetch_add(threadIdx.x == 0 ? p.done : (threadIdx.x == 1 ? p.abort_flag : p.q_tail), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t dn = (uint32_t)__builtin_amdgcn_readlane((int)v3, 0), ab = (uint32_t)__builtin_amdgcn_readlane((int)v3, 1),
                           tl = (uint32_t)__builtin_amdgcn_readlane((int)v3, 2);
            // the stale-counter question: the atomics above have returned; sc1 loads of the same (monotonic) words now
            uint32_t w3 = 0;
            if (threadIdx.x < 3 && threadIdx.x != 1) w3 = ld_sc1(threadIdx.x == 0 ? p.done : p.q_tail);
            const uint32_t dn_l = (uint32_t)__builtin_amdgcn_readlane((int)w3, 0), tl_l = (uint32_t)__builtin_amdgcn_readlane((int)w3, 2);
            if (threadIdx.x == 0) {
                count(p, 5, 1);
                count(p, 3, dn_l < dn ? 1 : 0);                       // "windows done" read stale by an sc1 load
                count(p, 4, (int32_t)(tl_l - tl) < 0 ? 1 : 0);        // the queue's tail read stale by an sc1 load
            }
            if (dn >= p.G || ab) return 0xffffffffu;
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (tl != seen_tail) { seen_tail = tl; t_push = now; }
            if (now - t_push > p.watchdog_ticks) {
                if (threadIdx.x == 0) __hip_atomic_fetch_add(p.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return 0xffffffffu;
            }
        }
        __builtin_amdgcn_s_sleep(16);
    }
}

__device__ void push_generation(const Probe& p, uint32_t g, uint32_t gen) {
    const uint32_t lane = threadIdx.x;
    unsigned long long* rec = p.grec + (size_t)g * kWords;
    if (lane < (uint32_t)kWords && lane != 1) st_sc1(rec + lane, tag(1, g, gen, lane));
    if (lane == 1) st_sc1((uint32_t*)(rec + 1), p.S); // the counter (only atomics change it afterwards)
    uint32_t base = 0;
    if (lane == 0) base = __hip_atomic_fetch_add(p.q_tail, p.S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wait_stores();
    base = uni(base);
    for (uint32_t i = lane; i < p.S; i += 64) {
        const uint32_t e = base + i;
        st_sc1(p.q + (e & p.q_mask), ((unsigned long long)((e >> p.q_shift) + 1u) << 32) | (gen << kSlotBits) | (g * p.S + i));
    }
}

__global__ __launch_bounds__(64) void probe_kernel(Probe p) {
    const uint32_t lane = threadIdx.x;
    for (;;) {
        const uint32_t cellv = pop(p);
        if (cellv == 0xffffffffu) break;
        const uint32_t slot = cellv & ((1u << kSlotBits) - 1u);
        const uint32_t gen = cellv >> kSlotBits; // the cell carries the generation it was pushed for
        const uint32_t g = slot / p.S;
        unsigned long long* rec = p.grec + (size_t)g * kWords;
        // 1. the group's record, written by the wave that pushed this generation: every word must be this generation's
        const unsigned long long r = lane < (uint32_t)kWords ? ld_sc1(rec + lane) : 0ull;
        {
            const bool bad = lane < (uint32_t)kWords && lane != 1 && r != tag(1, g, gen, lane);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(bad);
            if (lane == 0) count(p, 2, (unsigned long long)__builtin_popcountll(m));
        }
        // 2. this slot's payload of the previous generation
        unsigned long long* pl = p.payload + (size_t)slot * kWords;
        if (gen > 0) {
            const unsigned long long v = lane < (uint32_t)kWords ? ld_sc1(pl + lane) : 0ull;
            const bool bad = lane < (uint32_t)kWords && v != tag(2, slot, gen - 1, lane);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(bad);
            if (lane == 0) count(p, 0, (unsigned long long)__builtin_popcountll(m));
        }
        // 3. uneven work, then this generation's payload
        {
            const uint32_t n = (uint32_t)(tag(3, slot, gen, 0) >> 40) & p.sleep_mask;
            for (uint32_t i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
        }
        if (lane < (uint32_t)kWords) st_sc1(pl + lane, tag(2, slot, gen, lane));
        wait_stores();
        uint32_t left = 0;
        if (lane == 0) left = __hip_atomic_fetch_sub((uint32_t*)(rec + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        left = uni(left);
        if (lane == 0) count(p, 6, 1);
        if (left != 1u) continue;
        // 4. the last task of the generation: every payload of the group, every word
        {
            unsigned long long bad_n = 0;
            for (uint32_t e = lane; e < p.S * kWords; e += 64) {
                const uint32_t s2 = g * p.S + e / kWords, w = e % kWords;
                const unsigned long long v = ld_sc1(p.payload + (size_t)s2 * kWords + w);
                bad_n += v != tag(2, s2, gen, w) ? 1u : 0u;
            }
            count(p, 1, bad_n);
        }
        if (gen + 1 < p.gens) {
            push_generation(p, g, gen + 1);
        } else {
            wait_stores();
            uint32_t before = 0;
            if (lane == 0) before = __hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (uni(before) + 1u == p.G) { // end markers, one per wave of the launch
                const uint32_t n = gridDim.x;
                uint32_t base = 0;
                if (lane == 0) base = __hip_atomic_fetch_add(p.q_tail, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                base = uni(base);
                for (uint32_t i = lane; i < n; i += 64) {
                    const uint32_t e = base + i;
                    st_sc1(p.q + (e & p.q_mask), ((unsigned long long)((e >> p.q_shift) + 1u) << 32) | 0xffffffffull);
                }
            }
        }
    }
}

unsigned long long h_tag(uint32_t kind, uint32_t id, uint32_t gen, uint32_t word) {
    unsigned long long z = ((unsigned long long)kind << 60) ^ ((unsigned long long)id << 36) ^ ((unsigned long long)gen << 8) ^ word;
    z = (z ^ (z >> 31)) * 0x9E3779B97F4A7C15ull;
    return z ^ (z >> 29);
}

} // namespace

int main(int argc, char** argv) {
    const uint32_t G = 98, S = 61; // (G * S < 2^13)
    uint32_t gens = argc > 1 ? (uint32_t)atoi(argv[1]) : 400;
    if (gens < 1 || gens >= (1u << (31 - kSlotBits))) gens = 400;
    int n_cu = 256;
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("{\"what\": \"hand-off protocol of the window executor, every word checked (tools/ubench/handoff_probe.hip)\", \"groups\": %u, \"slots_per_group\": %u, "
           "\"generations\": %u, \"words_per_record\": %d, \"compute_units\": %d, \"runs\": [\n", G, S, gens, kWords, n_cu);
    bool first = true;
    for (int shared_line = 0; shared_line <= 1; ++shared_line)
        for (uint32_t per_cu : {8u, 4u, 1u})
            for (uint32_t sleep_mask : {63u, 0u}) {
                const uint32_t ns = G * S;
                uint32_t waves = (uint32_t)n_cu * per_cu;
                if (waves > ns) waves = ns;
                uint32_t q_cap = 256, q_shift = 8;
                while (q_cap < 4 * (ns + waves)) { q_cap *= 2; ++q_shift; }
                unsigned long long *d_q, *d_grec, *d_payload, *d_stats;
                uint32_t* d_ctl;
                CK(hipMalloc(&d_q, (size_t)q_cap * 8));
                CK(hipMalloc(&d_grec, (size_t)G * kWords * 8));
                CK(hipMalloc(&d_payload, (size_t)ns * kWords * 8));
                CK(hipMalloc(&d_stats, 16 * 8));
                CK(hipMalloc(&d_ctl, 4 * 128));
                std::vector<unsigned long long> queue(q_cap, 0ull), grec((size_t)G * kWords);
                for (uint32_t j = 0; j < ns; ++j) queue[j] = (1ull << 32) | j;
                for (uint32_t g = 0; g < G; ++g)
                    for (int w = 0; w < kWords; ++w) grec[(size_t)g * kWords + w] = w == 1 ? (unsigned long long)S : h_tag(1, g, 0, w);
                const uint32_t stride = shared_line ? 1u : 32u; // in 4-byte words: one 16-byte block, or a 128-byte line each
                std::vector<uint32_t> ctl(128, 0u);
                ctl[stride] = ns; // head 0, tail ns, done 0, abort 0
                // pageable host memory through hipMemcpyAsync, as rship_sync_exec initialises its state
                CK(hipMemcpyAsync(d_q, queue.data(), (size_t)q_cap * 8, hipMemcpyHostToDevice, 0));
                CK(hipMemcpyAsync(d_grec, grec.data(), grec.size() * 8, hipMemcpyHostToDevice, 0));
                CK(hipMemcpyAsync(d_ctl, ctl.data(), 512, hipMemcpyHostToDevice, 0));
                CK(hipMemsetAsync(d_stats, 0, 16 * 8, 0));
                CK(hipMemsetAsync(d_payload, 0, (size_t)ns * kWords * 8, 0));
                Probe p{};
                p.q = d_q; p.q_mask = q_cap - 1; p.q_shift = q_shift;
                p.q_head = d_ctl; p.q_tail = d_ctl + stride; p.done = d_ctl + 2 * stride; p.abort_flag = d_ctl + 3 * stride;
                p.grec = d_grec; p.payload = d_payload;
                p.G = G; p.S = S; p.gens = gens;
                p.stats = d_stats;
                p.watchdog_ticks = 200000000ull; // 2 s
                p.sleep_mask = sleep_mask;
                p.poll_mask = 15; // (the executor: 1023; more frequent here, to give a stale counter read every chance)
                hipEvent_t a, b;
                CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
                CK(hipEventRecord(a, 0));
                hipLaunchKernelGGL(probe_kernel, dim3(waves), dim3(64), 0, 0, p);
                CK(hipGetLastError());
                CK(hipEventRecord(b, 0));
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventElapsedTime(&ms, a, b));
                unsigned long long st[16];
                uint32_t h_ctl[128];
                CK(hipMemcpy(st, d_stats, sizeof(st), hipMemcpyDeviceToHost));
                CK(hipMemcpy(h_ctl, d_ctl, 512, hipMemcpyDeviceToHost));
                printf("%s {\"control_words\": \"%s\", \"workgroups_per_cu\": %u, \"waves\": %u, \"uneven_load\": %s, \"ms\": %.2f, \"tasks\": %llu, "
                       "\"words_checked\": %llu, \"stale_words_own_slot\": %llu, \"stale_words_group_gather\": %llu, \"stale_words_group_record\": %llu, "
                       "\"counter_polls\": %llu, \"stale_done_counter_sc1_reads\": %llu, \"stale_tail_counter_sc1_reads\": %llu, "
                       "\"windows_done\": %u, \"aborted\": %u, \"ring_cells\": %u, \"numbers_pushed\": %u}",
                       first ? "" : ",\n", shared_line ? "one 16-byte block" : "a 128-byte line each", per_cu, waves, sleep_mask ? "true" : "false", ms, st[6],
                       st[6] * (unsigned long long)(2 * kWords - 1) + (unsigned long long)G * gens * S * kWords, st[0], st[1], st[2], st[5], st[3], st[4],
                       h_ctl[2 * stride], h_ctl[3 * stride], q_cap, h_ctl[stride]);
                first = false;
                fflush(stdout);
                CK(hipFree(d_q)); CK(hipFree(d_grec)); CK(hipFree(d_payload)); CK(hipFree(d_stats)); CK(hipFree(d_ctl));
                CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
                if (h_ctl[3 * stride]) { printf("\n]}\n"); fprintf(stderr, "aborted (watchdog or an unidentifiable record): stopping\n"); return 1; }
            }
    printf("\n]}\n");
    return 0;
}
