// The compiled witness plan (vpbs_witness_plan: value slots, generator schedule, the split into an early and a late phase, the device
// schedule) and the host thread pools that share the wide dependency levels of a phase.  Shared by witness.hip (creation, host runs) and
// witness_device.hip (which replays plan.dev on the GPU).
#pragma once
#include "witness_gen.h"

namespace vpbs {

// spin for up to 150 us (the other side usually answers within microseconds), then sleep in short naps: a pool's workers wait
// through the stretches of a phase that only the calling thread or the chain lanes work on, and a container's CPU quota is spent by
// spinning threads as by working ones
template <class Pred> void spin_until(Pred ready) {
    std::chrono::steady_clock::time_point t0;
    for (unsigned i = 0; !ready(); ++i) {
        __builtin_ia32_pause();
        if ((i & 63) != 63) continue;
        const auto t = std::chrono::steady_clock::now();
        if (i == 63) t0 = t;
        else if (t - t0 > std::chrono::milliseconds(1)) std::this_thread::sleep_for(std::chrono::microseconds(100));   // a very long wait
        else if (t - t0 > std::chrono::microseconds(150)) std::this_thread::sleep_for(std::chrono::microseconds(20));  // a long wait: nap
    }
}

// The host threads that share the wide dependency levels of a split plan's phase.  They belong to the plan and live as long as it does:
// asleep on a condition variable between runs, spinning on a phase counter while a run is active (a run is hundreds of levels within a
// few milliseconds; waking a sleeping thread per level would cost more than the level).  One run at a time per pool (`busy`); a second
// concurrent run of the same phase finds the pool taken and proceeds on its own thread alone.
struct LevelPool {
    const unsigned threads;                       // including the calling thread (number 0)
    std::mutex busy;
    explicit LevelPool(unsigned n, const char* role = "vpbs-pool") : threads(std::max(1u, n)) {
        for (unsigned t = 1; t < threads; ++t)
            workers.emplace_back([this, t, role] {
                (void)pthread_setname_np(pthread_self(), role);   // CPU time by role: /proc/<pid>/task/*/comm
                work(t);
            });
        place_near_caller();
    }
    ~LevelPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        for (auto& w : workers) w.join();
        if (domain >= 0) domains().give_back(domain, threads);
    }
    void begin() {   // wake the workers for a run
        {
            std::lock_guard<std::mutex> lk(m);
            active.store(true, std::memory_order_release);
        }
        cv.notify_all();
    }
    void end() {
        std::lock_guard<std::mutex> lk(m);
        active.store(false, std::memory_order_release);
    }
    // A run may leave the workers spinning for the run that follows within a fraction of a millisecond (the last late stage of a chained step).
    // Whoever was promised that (a witness state) holds a TOKEN, not the pool: a run that takes the pool in between invalidates it (the
    // workers are that run's now), so a state freed or advanced from another thread can never put the workers of somebody else's run to
    // sleep; and the state shares ownership of the pool, so a plan that is re-split or freed first leaves nothing dangling.
    // Token and `active` change together under `m`: with the compare outside the lock a state could win it, a run could then
    // take the pool, void the token and begin(), and the state's end() -- late -- would put that run's workers to sleep under its share().
    unsigned long keep_awake_token() {
        static std::atomic<unsigned long> next{1};
        const unsigned long t = next.fetch_add(1, std::memory_order_relaxed);
        std::lock_guard<std::mutex> lk(m);
        awake_token = t;
        return t;
    }
    void taken_by_a_run() {
        std::lock_guard<std::mutex> lk(m);
        awake_token = 0;
    }
    void end_if_still_kept(unsigned long token) {
        std::lock_guard<std::mutex> lk(m);
        if (token && awake_token == token) {
            awake_token = 0;
            active.store(false, std::memory_order_release);
        }
    }
    unsigned long awake_token = 0;   // guarded by m
    // every thread of the pool (the caller as number 0) runs job(t); returns when all have finished
    template <class Job> void share(Job&& j) {
        job = std::ref(j);
        done.store(0, std::memory_order_relaxed);
        phase.fetch_add(1, std::memory_order_release);
        j(0u);
        spin_until([&] { return done.load(std::memory_order_acquire) == threads - 1; });
    }

    // the same without the caller: the workers run job(t), t = 1 .. threads - 1, while the caller does something else; finish() waits for them
    template <class Job> void start(Job&& j) {
        job = std::ref(j);
        done.store(0, std::memory_order_relaxed);
        phase.fetch_add(1, std::memory_order_release);
    }
    void finish() {
        spin_until([&] { return done.load(std::memory_order_acquire) == threads - 1; });
    }

  private:
    // The workers share slot values with the calling thread level after level: on a many-chiplet host they are kept on the CPUs of ONE
    // last-level-cache domain (Linux: cache/index3/shared_cpu_list) -- the caller's when it still has room for them, otherwise the least
    // taken domain of the same package (a process runs several pools: two phases per plan, one plan per concurrent chain; two pools spinning
    // on the same sixteen hardware threads cost more than the placement gains).  The caller's own affinity is left alone.
    // VPBS_POOL_PIN=0 turns the placement off.
    struct Domains {
        struct D {
            cpu_set_t set;
            unsigned cpus = 0, taken = 0;
            int package = -1;
        };
        std::mutex m;
        std::vector<D> all;
        std::vector<int> of_cpu;   // cpu -> index into all, -1 unknown
        static bool read_line(const char* fmt, int cpu, char* line, size_t len) {
            char path[160];
            std::snprintf(path, sizeof path, fmt, cpu);
            FILE* f = std::fopen(path, "r");
            if (!f) return false;
            const bool got = std::fgets(line, (int)len, f) != nullptr;
            std::fclose(f);
            return got;
        }
        Domains() {
            const long n = std::min<long>(sysconf(_SC_NPROCESSORS_CONF), CPU_SETSIZE);
            of_cpu.assign(n > 0 ? (size_t)n : 0, -1);
            for (int cpu = 0; cpu < (int)of_cpu.size(); ++cpu) {
                if (of_cpu[cpu] >= 0) continue;
                char line[512] = {0};
                if (!read_line("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu, line, sizeof line)) continue;
                D d;
                CPU_ZERO(&d.set);
                for (const char* q = line; *q;) {   // "0-7,128-135"
                    char* end = nullptr;
                    const long a = std::strtol(q, &end, 10);
                    if (end == q) break;
                    long b = a;
                    if (*end == '-') b = std::strtol(end + 1, &end, 10);
                    for (long c = a; c <= b && c < (long)of_cpu.size(); ++c) {
                        CPU_SET((int)c, &d.set);
                        ++d.cpus;
                        of_cpu[c] = (int)all.size();
                    }
                    if (*end != ',') break;
                    q = end + 1;
                }
                char pk[64] = {0};
                if (read_line("/sys/devices/system/cpu/cpu%d/topology/physical_package_id", cpu, pk, sizeof pk)) d.package = std::atoi(pk);
                if (d.cpus) all.push_back(d);
            }
        }
        // -> index of the domain that takes `n` more threads, or -1
        int take(unsigned n) {
            std::lock_guard<std::mutex> lk(m);
            const int cpu = sched_getcpu();
            const int home = cpu >= 0 && cpu < (int)of_cpu.size() ? of_cpu[cpu] : -1;
            if (home < 0) return -1;
            int best = -1;
            if (all[home].taken + n <= all[home].cpus) best = home;
            for (int pass = 0; pass < 2 && best < 0; ++pass)   // same package first
                for (int i = 0; i < (int)all.size(); ++i) {
                    if (all[i].cpus < n || (pass == 0 && all[i].package != all[home].package)) continue;
                    if (best < 0 || all[i].taken < all[best].taken) best = i;
                }
            if (best >= 0) all[best].taken += n;
            return best;
        }
        void give_back(int i, unsigned n) {
            std::lock_guard<std::mutex> lk(m);
            if (i >= 0) all[i].taken -= std::min(n, all[i].taken);
        }
    };
    static Domains& domains() {
        static Domains d;
        return d;
    }
    int domain = -1;
    void place_near_caller() {
        const char* e = std::getenv("VPBS_POOL_PIN");
        if ((e && std::atoi(e) == 0) || workers.empty()) return;
        domain = domains().take(threads);
        if (domain < 0) return;   // no cache topology to read, or no domain can hold the pool: the scheduler places the threads
        // inside what the process is allowed (a taskset / cpuset mask: the rank's CPU share) -- a pool must not leave it; when the domain and
        // the mask share fewer CPUs than the pool has threads, the scheduler places them (inside the mask, which the workers inherit)
        cpu_set_t allowed, both;
        CPU_ZERO(&both);
        if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
        CPU_AND(&both, &allowed, &domains().all[domain].set);
        if ((unsigned)CPU_COUNT(&both) < std::min(threads, (unsigned)CPU_COUNT(&allowed))) return;
        for (auto& w : workers) (void)pthread_setaffinity_np(w.native_handle(), sizeof(cpu_set_t), &both);
    }
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<u32> phase{0}, done{0};
    std::atomic<bool> active{false};
    bool quit = false;
    std::function<void(unsigned)> job;
    void work(unsigned t) {
        u32 seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return quit || active.load(std::memory_order_acquire); });
                if (quit) return;
            }
            for (;;) {
                spin_until([&] { return phase.load(std::memory_order_acquire) != seen || !active.load(std::memory_order_acquire); });
                if (phase.load(std::memory_order_acquire) == seen) break;   // the run is over: back to sleep
                ++seen;
                job(t);
                done.fetch_add(1, std::memory_order_release);
            }
        }
    }
};
}  // namespace vpbs

struct vpbs_witness_plan {
    using u32 = vpbs::u32;
    using u64 = vpbs::u64;
    size_t n = 0, total = 0, n_slots = 0;
    unsigned max_consts = 0;
    std::vector<vpbs_gate> gates;
    std::vector<u32> row_gate;                  // [n]
    std::vector<u64> consts;                    // [n][max_consts]: the gate constants of every row
    std::vector<u32> row_off, row_slots;        // row -> offset into row_slots: the slot of every wire of a row that owns generators
    // PoseidonGate rows: the slot of wire 29 when wires 29 .. 134 (the S-box inputs: wires nothing else refers to) sit on 106 consecutive slots of
    // their own -- the eight-lane generator then stores them as one range per row instead of 106 checked single stores; NONE otherwise
    std::vector<u32> poseidon_private_base;     // [n]
    struct Gadget {
        unsigned kind, p0;
        u32 at, n_in, n_out;                    // gadget_slots / gadget_pos [at, at + n_in) inputs, then n_out outputs
    };
    std::vector<Gadget> gadgets;
    std::vector<u32> gadget_slots, gadget_pos;
    struct Step {
        u32 row, sub;                           // row == NO_ROW: gadget number `sub`
    };
    std::vector<Step> schedule;
    std::vector<u32> preset_slot, preset_pos;
    std::vector<u32> out_pos, out_slot;         // every position that carries a slot, ascending (full_witness)
    // ---- two-phase runs (vpbs_witness_plan_split): some presets arrive late (the previous proof of an IVC step); a generator is LATE when
    // anything it reads is, everything else can run before the late values exist
    bool is_split = false;
    std::vector<uint8_t> preset_late, step_late;
    std::vector<u32> late_out;                  // indices into out_pos / out_slot whose slot is late, ordered by (stage the value exists from, slot)
    std::vector<u32> late_out_stage_off;        // [n_stages + 1]: late_out entries of stage k = [off[k - 1], off[k])
    // the late generators by dependency level (a generator of level L reads only what levels < L wrote, and two generators of one level
    // never write the same slot: a second writer of a slot is placed above the first and compares): the wide levels -- the 28 FRI queries
    // of an in-circuit verifier are independent of each other -- are run by several host threads
    // Both phases hold their generators by dependency level like that (the early phase of the step circuit: the NTT / external-product
    // arithmetic is wide, the hash chains are one PoseidonGate row per level).
    struct Phase {
        std::vector<u32> order, level_off;      // schedule indices sorted by level; level l = order[level_off[l], level_off[l + 1])
        std::vector<u32> cost;                  // [order.size() + 1] prefix sums of the generators' estimated cost (units of 10 ns)
        // STRANDS (stages of a staged late phase): when the stage's generators fall into many independent components -- the 28 FRI queries
        // of an in-circuit verifier share nothing once the transcript (an earlier stage) has produced their indices -- the components are
        // packed into `strand_threads` bins of about equal cost and every thread of the pool runs ONE bin from its first level to its last
        // with no barrier at all: what a generator reads was written by the same thread (or by an earlier stage).  strand t =
        // strand_steps[strand_off[t], strand_off[t + 1]), sorted by level; strand_level: the level of every entry (rows of one level are
        // independent: the PoseidonGate rows among them go eight at a time).  Empty: the stage is one big component, it runs level by level.
        unsigned strand_threads = 0;
        std::vector<u32> strand_steps, strand_level, strand_off;
    };
    Phase phase[2];                             // [0] early, [1] late
    // The late presets may arrive in STAGES (vpbs_witness_plan_split: late[i] = 1, 2, ...: the sections of a proof become final at
    // different moments of the prover's run -- caps and openings before the FRI stage, the query rounds at its very end): a late
    // generator belongs to the highest stage among what it reads, and late_stage[k - 1] holds the generators of stage k by dependency
    // level, so that stage k can run as soon as its presets exist, while the values of the stages above are still being computed.
    unsigned n_stages = 1;
    std::vector<std::vector<u32>> stage_presets;   // [n_stages]: indices of the presets that arrive in stage k + 1
    std::vector<uint8_t> step_stage;            // per scheduled step: 0 early, otherwise its late stage
    std::vector<Phase> late_stage;              // [n_stages] (only filled when n_stages > 1; one stage = phase[1])
    // Chain lanes: PoseidonGate rows whose results nothing but other such rows reads inside the phase (the hash chains over the GGSW and
    // over the public inputs: thousands of permutations, each needing the one before).  They are taken out of the levels and run on
    // threads of their own next to them, in schedule order, waiting on the set flag of a value another lane or a level still has to produce.
    struct Lane {
        std::vector<u32> steps;                 // schedule indices, ascending
        std::vector<u32> wait_off, wait;        // step k waits for slots wait[wait_off[k], wait_off[k + 1])
    };
    std::vector<Lane> lanes[2];
    std::vector<u32> lane_steps_sorted[2];      // all lane steps of a phase, ascending (a run without threads for the lanes)
    mutable std::shared_ptr<vpbs::LevelPool> pool[2];      // shared with the witness states a run left the workers spinning for
    mutable std::unique_ptr<vpbs::LevelPool> lane_pool[2];
    mutable std::mutex pool_mutex;              // creation of the pools (first run of a phase)

    // ---- the same schedule by dependency level, for the device (vpbs_witness_device_*): every generator of level L only reads slots
    // written at levels < L; within a level the operations are grouped by kind.  CHECK in an output slot index: the slot already has a
    // writer earlier in the schedule (a preset, or a generator whose output is copy-constrained to this one), so the operation
    // compares instead of writing ("set twice with different values") -- the order is static, so no per-slot flag is needed at run time.
    static constexpr u32 CHECK = 0x80000000u;
    struct ArithOp {
        u32 x, y, z, out;
        u64 c0, c1;
    };
    struct ConstOp {
        u32 out, pad;
        u64 value;
    };
    struct BitsOp {                             // out[k] = (in >> (k * bits)) & mask, k < n_out; the rest must be zero
        u32 in, out_at, n_out, bits;            // BaseSumGate<2> generator, WireSplitGenerator
    };
    struct MiscOp {                             // gadget generators that are not bit splits
        u32 kind, p0, at, n_in, n_out, pad;
    };
    struct RowOp {                              // any other gate generator: gen_run on the row's slot table
        u32 row, sub;
    };
    struct DeviceSchedule {
        bool supported = false;
        std::string unsupported;                // why not (a gate or generator kind without a device form)
        u32 n_levels = 0;
        std::vector<ArithOp> arith;
        std::vector<ConstOp> consts;            // no inputs: written before level 1
        std::vector<BitsOp> bits;
        std::vector<u32> poseidon;              // offsets into row_slots (135 slots per PoseidonGate row)
        std::vector<MiscOp> misc;
        std::vector<RowOp> rowops;
        std::vector<u32> arith_off, bits_off, poseidon_off, misc_off, rowops_off;   // [n_levels + 2]: operations of level L = [off[L], off[L + 1])
        std::vector<u32> aux;                   // slot lists of BitsOp outputs and MiscOp inputs / outputs
        std::vector<u32> row_slots;             // copy of row_slots with CHECK marks on outputs that compare
        std::vector<u32> preset_slot;           // with CHECK marks (a target preset twice)
    } dev;
    // After vpbs_witness_plan_split: the schedule of the EARLY phase alone (late steps left out, late presets routed to the scratch slot
    // n_slots), for a host that generates the early part of a chain's witnesses on the device in batches and runs only the late phase --
    // the part that needs the previous proof -- on its CPUs; and the early-known slots that late phase touches (reads, or writes as a
    // comparer), which such a host reads back per instance to seed the late phase's state.
    DeviceSchedule dev_early;
    // ... and the LATE phase alone: the late steps over a value array in which the early phase's results already sit (the early-known
    // classes count as set: a late step that writes one compares), early presets routed to the scratch slot
    DeviceSchedule dev_late;
    // ... and that late phase STAGE BY STAGE (a split with stage numbers): dev_late_stage[k - 1] holds the steps and presets of stage k over
    // a value array in which the early phase and the stages before it have left their results, so that a host can queue a stage's walk
    // as soon as that section of the previous proof is final -- what the host's own late phase does (LateAhead in ivc.hip).  Their
    // row-slot tables are one shared array (late_row_slots: the plan's with every stage's CHECK marks; a wire is written by one step).
    std::vector<DeviceSchedule> dev_late_stage;
    std::vector<u32> late_row_slots;
    std::vector<u32> late_in_slots;
    // runs [first, first + count) of consecutive value slots that only the late phase writes: a fresh state's pages under them are touched
    // by whoever creates the state (the early thread, the stager) instead of faulting in one by one on the critical path of the late phase
    std::vector<std::pair<u32, u32>> late_slot_runs;
    // the packed late wires as copies of slot ranges: entry (k, slot, len) = late_out[k + i] holds slot + i, i < len; ordered by k, cut at the
    // stage boundaries.  A PoseidonGate row's 110 private wires are one run: packing is a few thousand range copies, not 618 000 lookups
    struct PackRun {
        u32 k, slot, len;
    };
    std::vector<PackRun> pack_runs;
    // what every scheduled step writes (slots; wire index for row steps, NONE for gadget outputs): kept for building dev_early
    std::vector<u32> step_out, step_out_w, step_out_off;
};
