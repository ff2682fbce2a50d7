// Witness generation for circuits made of the supported gates: on the host (vpbs_generate_witness, vpbs_witness_plan_*) and, for a
// batch of PartialWitnesses of one circuit, on the device (vpbs_witness_device_*: witness_device.hip).  Generators: witness_gen.h; plan + pools: witness_plan.h.
// Replaces plonky2 0.2.0 iop/generator.rs `generate_partial_witness` (run every generator whose watched targets are set,
// propagate through the copy-constraint partition, repeat), iop/witness.rs `PartitionWitness::full_witness`, the gates' own
// `SimpleGenerator::run_once` implementations (gates/*.rs), plonk/permutation_argument.rs `WirePartition::get_sigma_polys` and
// the selector columns of gates/selectors.rs -- the first stage of prove() reached from
// /root/reference/src/vtfhe/ivc_based_vpbs.rs:302,333,364.  The gadget-level generators of the
// reference's own circuit builder (vec_arithmetic, the recursive verifier) are outside: the caller presets their targets.
// Restated from the published crate: parity unpinned; checked against an independent Python restatement and against the
// gate constraints (every generated row must satisfy them).
#include "witness_plan.h"

namespace vpbs {
namespace {
constexpr u32 NO_ROW = 0xFFFFFFFFu, NONE = 0xFFFFFFFFu;

void report(char* err, size_t err_len, const std::string& m) {
    if (err && err_len) {
        std::strncpy(err, m.c_str(), err_len - 1);
        err[err_len - 1] = 0;
    }
}

// plan creation: generators "run" on readiness flags only (every read returns 1, which no generator rejects)
struct FlagRow {
    std::vector<uint8_t>& ready;
    const u32* rs;
    std::vector<u32>& written;    // the slots this generator sets, in order
    std::vector<u32>& written_w;  // ... and their wire indices
    u64 get(unsigned) { return 1; }
    void set(unsigned w, u64) {
        ready[rs[w]] = 1;
        written.push_back(rs[w]);
        written_w.push_back(w);
    }
    void fail(const char* m) { throw GenError{m}; }
};

// a plain array that is not zeroed unless asked: a slot state is tens of MB, and nothing reads a value whose set flag is clear
template <class T> struct Buf {
    std::unique_ptr<T[]> p;
    size_t n = 0;
    Buf(size_t count, bool zero) : p(zero ? new T[count]() : new T[count]), n(count) {}
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    T* data() { return p.get(); }
    size_t size() const { return n; }
};

struct SlotState {
    Buf<u64> val;
    Buf<uint8_t> is_set;
    size_t n;
    std::string error;        // written under report_mutex(); other threads look at has_error while generators may still be running
    uint8_t has_error = 0;    // and read the text through error_text(), or directly once every thread of the run has finished
    static std::mutex& report_mutex() {
        static std::mutex m;   // the phases of a split plan run generators on several threads
        return m;
    }
    bool failed() const { return __atomic_load_n(&has_error, __ATOMIC_ACQUIRE) != 0; }
    std::string error_text() const {
        std::lock_guard<std::mutex> lock(report_mutex());
        return error;
    }
    void set(u32 slot, u64 v, u32 pos) {
        if (v >= gl::P) v -= gl::P;
        if (is_set[slot]) {
            // a second writer only compares: in a split plan another thread may be reading this value right now (a generator of the same
            // level, a chain lane), so it is not stored again
            if (val[slot] != v) {
                std::lock_guard<std::mutex> lock(report_mutex());
                if (error.empty())
                    error = "partition containing wire (column " + std::to_string(pos / n) + ", row " + std::to_string(pos % n) +
                            ") was set twice with different values";
                __atomic_store_n(&has_error, (uint8_t)1, __ATOMIC_RELEASE);
            }
            return;
        }
        val[slot] = v;
        __atomic_store_n(&is_set[slot], (uint8_t)1, __ATOMIC_RELEASE);   // a chain lane of a split plan may be waiting for this slot
    }
    bool ready(u32 slot) const { return __atomic_load_n(&is_set[slot], __ATOMIC_ACQUIRE) != 0; }
};

struct SlotRow {
    SlotState& s;
    const u32* rs;
    u32 row;
    u64 get(unsigned w) { return s.val[rs[w]]; }
    void set(unsigned w, u64 v) { s.set(rs[w], v, (u32)(w * s.n + row)); }
    void fail(const char* m) { throw GenError{m}; }
};

// Levels and typed operation lists for the device (see vpbs_witness_plan::DeviceSchedule).  step_out: the slots every step of the
// schedule writes, in the order the generator sets them.
// skip_step / skip_preset (nullable): the steps and presets of the late phase, left out of the early-only schedule (such a preset is routed
// to the scratch slot n_slots, which nothing reads).
// pre_set (nullable, per slot): classes that hold a value before the schedule starts (the late-only schedule runs on the early phase's values).
void build_device_schedule(vpbs_witness_plan& p, vpbs_witness_plan::DeviceSchedule& d, const std::vector<u32>& step_out,
                           const std::vector<u32>& step_out_w, const std::vector<u32>& step_out_off, const uint8_t* skip_step = nullptr,
                           const uint8_t* skip_preset = nullptr, const uint8_t* pre_set = nullptr) {
    using Plan = vpbs_witness_plan;
    d = Plan::DeviceSchedule{};
    constexpr u32 CHECK = Plan::CHECK, UNSET = 0xFFFFFFFFu;
    if (p.n_slots + 1 >= CHECK) {
        d.unsupported = "too many value slots";
        return;
    }
    std::vector<u32> level(p.n_slots, UNSET);  // the level at which a slot gets its value
    if (pre_set)
        for (size_t sl = 0; sl < p.n_slots; ++sl)
            if (pre_set[sl]) level[sl] = 0;
    d.preset_slot = p.preset_slot;
    for (size_t i = 0; i < d.preset_slot.size(); ++i) {
        u32& s = d.preset_slot[i];
        if (skip_preset && skip_preset[i]) {
            s = (u32)p.n_slots;
            continue;
        }
        if (level[s] != UNSET) s |= CHECK;
        else level[s] = 0;
    }
    d.row_slots = p.row_slots;
    const unsigned mc = std::max(1u, p.max_consts);
    struct Tagged {
        u32 level, kind, index;
    };
    std::vector<Plan::ArithOp> arith;
    std::vector<Plan::BitsOp> bits;
    std::vector<u32> poseidon;
    std::vector<Plan::MiscOp> misc;
    std::vector<Plan::RowOp> rowops;
    std::vector<u32> l_arith, l_bits, l_pos, l_misc, l_row;
    std::vector<unsigned> deps;
    u32 max_level = 0;
    for (size_t i = 0; i < p.schedule.size(); ++i) {
        if (skip_step && skip_step[i]) continue;
        const auto& st = p.schedule[i];
        const u32* outs = step_out.data() + step_out_off[i];
        const u32 n_outs = step_out_off[i + 1] - step_out_off[i];
        u32 lvl = 1;
        auto need = [&](u32 slot) {
            if (level[slot] != UNSET) lvl = std::max(lvl, level[slot] + 1);
        };
        // an output that already has a writer is compared, after that writer
        auto mark = [&](u32 slot) -> u32 {
            if (level[slot] != UNSET) return slot | CHECK;
            level[slot] = lvl;
            return slot;
        };
        if (st.row == NO_ROW) {
            const auto& gg = p.gadgets[st.sub];
            const u32* gs = p.gadget_slots.data() + gg.at;
            for (unsigned k = 0; k < gg.n_in; ++k) need(gs[k]);
            for (unsigned k = 0; k < gg.n_out; ++k) need(gs[gg.n_in + k]);
            if (gg.kind == VPBS_GEN_WIRE_SPLIT) {
                Plan::BitsOp op{gs[0], (u32)d.aux.size(), gg.n_out, gg.p0};
                for (unsigned k = 0; k < gg.n_out; ++k) d.aux.push_back(mark(gs[gg.n_in + k]));
                bits.push_back(op);
                l_bits.push_back(lvl);
            } else {
                Plan::MiscOp op{gg.kind, gg.p0, (u32)d.aux.size(), gg.n_in, gg.n_out, 0};
                for (unsigned k = 0; k < gg.n_in; ++k) d.aux.push_back(gs[k]);
                for (unsigned k = 0; k < gg.n_out; ++k) d.aux.push_back(mark(gs[gg.n_in + k]));
                misc.push_back(op);
                l_misc.push_back(lvl);
            }
        } else {
            const vpbs_gate& g = p.gates[p.row_gate[st.row]];
            const u32 off = p.row_off[st.row];
            const u32* rs = p.row_slots.data() + off;
            const u64* c = p.consts.data() + (size_t)st.row * mc;
            gen_deps(g, st.sub, deps);
            for (unsigned w : deps) need(rs[w]);
            for (u32 k = 0; k < n_outs; ++k) need(outs[k]);
            switch (g.kind) {
                case VPBS_GATE_CONSTANT:
                    if (level[rs[st.sub]] != UNSET) {  // a constant wire that is also preset: compare at level 1
                        misc.push_back(Plan::MiscOp{0xC0u, 0, (u32)d.aux.size(), 0, 1, 0});
                        d.aux.push_back(rs[st.sub] | CHECK);
                        d.aux.push_back((u32)c[st.sub]);
                        d.aux.push_back((u32)(c[st.sub] >> 32));
                        l_misc.push_back(lvl);
                    } else {
                        level[rs[st.sub]] = 0;
                        d.consts.push_back(Plan::ConstOp{rs[st.sub], 0, c[st.sub]});
                    }
                    break;
                case VPBS_GATE_ARITHMETIC:
                    arith.push_back(Plan::ArithOp{rs[4 * st.sub], rs[4 * st.sub + 1], rs[4 * st.sub + 2], mark(rs[4 * st.sub + 3]), c[0], c[1]});
                    l_arith.push_back(lvl);
                    break;
                case VPBS_GATE_BASE_SUM: {
                    if (g.p1 != 2) {  // other bases: the generic row operation
                        const u32* outs_w = step_out_w.data() + step_out_off[i];
                        for (u32 k = 0; k < n_outs; ++k) d.row_slots[off + outs_w[k]] = mark(outs[k]);
                        rowops.push_back(Plan::RowOp{st.row, st.sub});
                        l_row.push_back(lvl);
                        break;
                    }
                    Plan::BitsOp op{rs[0], (u32)d.aux.size(), g.p0, 1};
                    for (unsigned k = 0; k < g.p0; ++k) d.aux.push_back(mark(rs[1 + k]));
                    bits.push_back(op);
                    l_bits.push_back(lvl);
                    break;
                }
                case VPBS_GATE_POSEIDON:
                default: {
                    // the generator reads and writes through the row's slot table; outputs that compare carry the mark there
                    const u32* outs_w = step_out_w.data() + step_out_off[i];
                    for (u32 k = 0; k < n_outs; ++k) d.row_slots[off + outs_w[k]] = mark(outs[k]);
                    if (g.kind == VPBS_GATE_POSEIDON) {
                        poseidon.push_back(off);
                        l_pos.push_back(lvl);
                    } else {
                        rowops.push_back(Plan::RowOp{st.row, st.sub});
                        l_row.push_back(lvl);
                    }
                    break;
                }
            }
        }
        max_level = std::max(max_level, lvl);
    }
    d.n_levels = max_level;
    // counting sort of every kind by level
    auto by_level = [&](auto& ops, const std::vector<u32>& lv, auto& sorted, std::vector<u32>& off) {
        off.assign(max_level + 2, 0);
        for (u32 l : lv) ++off[l + 1];
        for (u32 l = 0; l <= max_level; ++l) off[l + 1] += off[l];
        sorted.resize(ops.size());
        std::vector<u32> at(off.begin(), off.end() - 1);
        for (size_t i = 0; i < ops.size(); ++i) sorted[at[lv[i]]++] = ops[i];
    };
    by_level(arith, l_arith, d.arith, d.arith_off);
    by_level(bits, l_bits, d.bits, d.bits_off);
    by_level(poseidon, l_pos, d.poseidon, d.poseidon_off);
    by_level(misc, l_misc, d.misc, d.misc_off);
    by_level(rowops, l_row, d.rowops, d.rowops_off);
    d.supported = true;
}

int plan_create(const vpbs_circuit* c, const u32* preset_pos, size_t n_preset, vpbs_witness_plan** out, std::string& err) {
    auto fail = [&](const std::string& m) {
        err = m;
        return VPBS_ERR_INVALID;
    };
    if (!check_circuit(c) || !out || (n_preset && !preset_pos)) return fail("malformed circuit description");
    const size_t n = (size_t)1 << c->log_n, total = (size_t)c->n_wires * n;
    unsigned max_consts = 0;
    for (unsigned i = 0; i < c->n_gates; ++i) max_consts = std::max(max_consts, c->gates[i].num_constants);
    if (c->num_selectors + max_consts > c->n_constants_cols || (max_consts && !c->constants)) return fail("constants columns missing");
    for (size_t i = 0; i < n_preset; ++i)
        if (preset_pos[i] >= total) return fail("preset position out of range");
    auto plan = std::make_unique<vpbs_witness_plan>();
    vpbs_witness_plan& p = *plan;
    p.n = n;
    p.total = total;
    p.max_consts = max_consts;
    p.gates.assign(c->gates, c->gates + c->n_gates);
    p.row_gate.assign(c->row_gate, c->row_gate + n);
    p.consts.assign((size_t)std::max(1u, max_consts) * n, 0);
    for (size_t r = 0; r < n; ++r)
        for (unsigned k = 0; k < p.gates[p.row_gate[r]].num_constants; ++k)
            p.consts[r * std::max(1u, max_consts) + k] = c->constants[(size_t)(c->num_selectors + k) * n + r];
    // copy-constraint classes; advice wires are singletons
    Partition part(total);
    for (size_t i = 0; i < c->n_copies; ++i) part.merge(c->copies[2 * i], c->copies[2 * i + 1]);
    std::vector<uint8_t> touched(total, 0);
    for (size_t r = 0; r < n; ++r)
        for (unsigned w = 0; w < p.gates[p.row_gate[r]].num_wires; ++w) touched[(size_t)w * n + r] = 1;
    for (size_t i = 0; i < 2 * c->n_copies; ++i) touched[c->copies[i]] = 1;
    for (size_t i = 0; i < n_preset; ++i) touched[preset_pos[i]] = 1;
    for (size_t i = 0; i < c->n_generators; ++i) {
        for (unsigned k = 0; k < c->generators[i].n_in; ++k) touched[c->generators[i].in[k]] = 1;
        for (unsigned k = 0; k < c->generators[i].n_out; ++k) touched[c->generators[i].out[k]] = 1;
    }
    std::vector<u32> slot_of_root(total, NONE);
    u32 n_slots = 0;
    for (size_t pos = 0; pos < total; ++pos) {
        if (!touched[pos]) continue;
        u32& s = slot_of_root[part.find((u32)pos)];
        if (s == NONE) s = n_slots++;
        p.out_pos.push_back((u32)pos);
        p.out_slot.push_back(s);
    }
    p.n_slots = n_slots;
    auto slot_at = [&](u32 pos) { return slot_of_root[part.find(pos)]; };
    p.row_off.assign(n, NONE);
    for (size_t r = 0; r < n; ++r) {
        const vpbs_gate& g = p.gates[p.row_gate[r]];
        if (gen_count(g) == 0) continue;
        p.row_off[r] = (u32)p.row_slots.size();
        for (unsigned w = 0; w < g.num_wires; ++w) p.row_slots.push_back(slot_at((u32)(w * n + r)));
    }
    for (size_t i = 0; i < c->n_generators; ++i) {
        const vpbs_generator& g = c->generators[i];
        p.gadgets.push_back({g.kind, g.p0, (u32)p.gadget_slots.size(), g.n_in, g.n_out});
        for (unsigned k = 0; k < g.n_in + g.n_out; ++k) {
            const u32 pos = k < g.n_in ? g.in[k] : g.out[k - g.n_in];
            p.gadget_slots.push_back(slot_at(pos));
            p.gadget_pos.push_back(pos);
        }
    }
    p.preset_pos.assign(preset_pos, preset_pos + n_preset);
    for (size_t i = 0; i < n_preset; ++i) p.preset_slot.push_back(slot_at(preset_pos[i]));
    // generate_partial_witness on readiness flags: run whatever is ready until nothing changes
    std::vector<uint8_t> ready(n_slots, 0);
    for (u32 s : p.preset_slot) ready[s] = 1;
    std::vector<vpbs_witness_plan::Step> pending, later;
    for (size_t r = 0; r < n; ++r)
        for (unsigned sub = 0; sub < gen_count(p.gates[p.row_gate[r]]); ++sub) pending.push_back({(u32)r, sub});
    for (size_t i = 0; i < p.gadgets.size(); ++i) pending.push_back({NO_ROW, (u32)i});
    std::vector<unsigned> deps;
    std::vector<u32> step_out, step_out_off{0};  // the slots every scheduled step writes (old numbering; renumbered below)
    std::vector<u32> step_out_w;                 // the wire index of each (row steps; NONE for gadget outputs)
    while (!pending.empty()) {
        later.clear();
        for (const auto& st : pending) {
            bool ok = true;
            if (st.row == NO_ROW) {
                const auto& gg = p.gadgets[st.sub];
                for (unsigned k = 0; k < gg.n_in && ok; ++k) ok = ready[p.gadget_slots[gg.at + k]];
                if (ok)
                    for (unsigned k = 0; k < gg.n_out; ++k) {
                        ready[p.gadget_slots[gg.at + gg.n_in + k]] = 1;
                        step_out.push_back(p.gadget_slots[gg.at + gg.n_in + k]);
                        step_out_w.push_back(NONE);
                    }
            } else {
                const vpbs_gate& g = p.gates[p.row_gate[st.row]];
                const u32* rs = p.row_slots.data() + p.row_off[st.row];
                gen_deps(g, st.sub, deps);
                for (unsigned w : deps)
                    if (!ready[rs[w]]) {
                        ok = false;
                        break;
                    }
                if (ok) {
                    FlagRow fr{ready, rs, step_out, step_out_w};
                    try {
                        gen_run(g, st.sub, p.consts.data() + (size_t)st.row * std::max(1u, max_consts), fr, tables_of(g));
                    } catch (const GenError& e) {
                        return fail(e.what + " (row " + std::to_string(st.row) + ")");
                    }
                }
            }
            if (ok) {
                p.schedule.push_back(st);
                step_out_off.push_back((u32)step_out.size());
            } else {
                later.push_back(st);
            }
        }
        if (later.size() == pending.size())  // generate_partial_witness: assert_eq!(remaining_generators, 0, "{} generators weren't run")
            return fail(std::to_string(later.size()) + " generators weren't run (first: " +
                        (later[0].row == NO_ROW ? "gadget generator " + std::to_string(later[0].sub) : "row " + std::to_string(later[0].row)) + ")");
        pending.swap(later);
    }
    // renumber the slots in the order the run first touches them (presets, then the schedule): values produced close in time
    // sit close in memory, instead of one cache line per wire column
    {
        std::vector<u32> renum(n_slots, NONE);
        u32 next = 0;
        auto visit = [&](u32& s) {
            if (renum[s] == NONE) renum[s] = next++;
        };
        for (u32& s : p.preset_slot) visit(s);
        std::vector<uint8_t> row_seen(n, 0);
        for (const auto& st : p.schedule) {
            if (st.row == NO_ROW) {
                const auto& gg = p.gadgets[st.sub];
                for (unsigned k = 0; k < gg.n_in + gg.n_out; ++k) visit(p.gadget_slots[gg.at + k]);
            } else if (!row_seen[st.row]) {
                row_seen[st.row] = 1;
                const unsigned nw = p.gates[p.row_gate[st.row]].num_wires;
                for (unsigned w = 0; w < nw; ++w) visit(p.row_slots[p.row_off[st.row] + w]);
            }
        }
        for (u32 s = 0; s < n_slots; ++s) visit(s);
        for (auto* v : {&p.preset_slot, &p.row_slots, &p.gadget_slots, &p.out_slot, &step_out})
            for (u32& s : *v) s = renum[s];
    }
    {   // PoseidonGate rows whose private wires own a run of consecutive slots (vpbs_witness_plan::poseidon_private_base)
        std::vector<uint8_t> refs(n_slots, 0);   // positions per class, saturating; anything a gadget or a preset refers to counts as shared
        for (u32 sl : p.out_slot)
            if (refs[sl] < 2) ++refs[sl];
        for (u32 sl : p.gadget_slots) refs[sl] = 2;
        for (u32 sl : p.preset_slot) refs[sl] = 2;
        p.poseidon_private_base.assign(n, NONE);
        for (size_t r = 0; r < n; ++r) {
            if (p.row_off[r] == NONE || p.gates[p.row_gate[r]].kind != VPBS_GATE_POSEIDON || p.gates[p.row_gate[r]].num_wires < 135) continue;
            const u32* rs = p.row_slots.data() + p.row_off[r];
            bool ok = true;
            for (unsigned w = 29; w < 135 && ok; ++w) ok = rs[w] == rs[29] + (w - 29) && refs[rs[w]] == 1;
            if (ok) p.poseidon_private_base[r] = rs[29];
        }
        if (std::getenv("VPBS_TRACE_WITNESS")) {
            size_t rows = 0, with_range = 0;
            for (size_t r = 0; r < n; ++r)
                if (p.row_off[r] != NONE && p.gates[p.row_gate[r]].kind == VPBS_GATE_POSEIDON) {
                    ++rows;
                    with_range += p.poseidon_private_base[r] != NONE;
                }
            std::fprintf(stderr, "[witness plan] %zu of %zu PoseidonGate rows keep their private wires on a slot range of their own\n", with_range, rows);
        }
    }
    build_device_schedule(p, p.dev, step_out, step_out_w, step_out_off);
    p.step_out = std::move(step_out);
    p.step_out_w = std::move(step_out_w);
    p.step_out_off = std::move(step_out_off);
    *out = plan.release();
    return VPBS_OK;
}

void run_gadget(const vpbs_witness_plan& p, const vpbs_witness_plan::Gadget& g, SlotState& s) {
    const u32 *in = p.gadget_slots.data() + g.at, *out = in + g.n_in, *out_pos = p.gadget_pos.data() + g.at + g.n_in;
    switch (g.kind) {
        case VPBS_GEN_EQUALITY: {
            const u64 x = s.val[in[0]], y = s.val[in[1]];
            s.set(out[0], x == y ? 1 : 0, out_pos[0]);
            s.set(out[1], x == y ? 0 : gl::inv(gl::sub(x, y)), out_pos[1]);
            break;
        }
        case VPBS_GEN_BASE_SUM: {
            u64 sum = 0;
            for (unsigned k = g.n_in; k-- > 0;) sum = gl::add(gl::mul(sum, g.p0), s.val[in[k]]);
            s.set(out[0], sum, out_pos[0]);
            break;
        }
        case VPBS_GEN_WIRE_SPLIT: {
            u64 x = s.val[in[0]];
            for (unsigned k = 0; k < g.n_out; ++k) {
                s.set(out[k], x & (((u64)1 << g.p0) - 1), out_pos[k]);
                x >>= g.p0;
            }
            if (x != 0) throw GenError{"WireSplitGenerator: integer too large to fit in the given number of limbs"};
            break;
        }
        case VPBS_GEN_QUOTIENT_EXT: {
            const A num{s.val[in[0]], s.val[in[1]]}, den{s.val[in[2]], s.val[in[3]]};
            if (den.a == 0 && den.b == 0) throw GenError{"QuotientGeneratorExtension: division by zero"};
            // 1 / (a + b X) = (a - b X) / (a^2 - 7 b^2)   (X^2 = 7)
            const u64 norm = gl::sub(gl::mul(den.a, den.a), gl::mul(7, gl::mul(den.b, den.b)));
            const A q = gates::scale(num * A{den.a, gl::neg(den.b)}, gl::inv(norm));
            s.set(out[0], q.a, out_pos[0]);
            s.set(out[1], q.b, out_pos[1]);
            break;
        }
        case VPBS_GEN_COPY: s.set(out[0], s.val[in[0]], out_pos[0]); break;
        case VPBS_GEN_LOW_HIGH: {
            const u64 x = s.val[in[0]];
            s.set(out[0], x & (((u64)1 << g.p0) - 1), out_pos[0]);
            s.set(out[1], x >> g.p0, out_pos[1]);
            break;
        }
        default: break;
    }
}

int plan_run(const vpbs_witness_plan* pp, const u64* preset_val, unsigned threads, u64* wires_out, std::string& err) {
    if (!pp || !wires_out || (!pp->preset_slot.empty() && !preset_val)) {
        err = "malformed arguments";
        return VPBS_ERR_INVALID;
    }
    const vpbs_witness_plan& p = *pp;
    const bool trace = std::getenv("VPBS_TRACE_WITNESS") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto t1 = std::chrono::steady_clock::now();
        if (trace) std::fprintf(stderr, "[witness] %-12s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    SlotState s{Buf<u64>(p.n_slots, true), Buf<uint8_t>(p.n_slots, true), p.n, {}};
    for (size_t i = 0; i < p.preset_slot.size(); ++i) s.set(p.preset_slot[i], preset_val[i], p.preset_pos[i]);
    if (!s.error.empty()) {
        err = s.error;
        return VPBS_ERR_INVALID;
    }
    lap("presets");
    const unsigned mc = std::max(1u, p.max_consts);
    for (const auto& st : p.schedule) {
        try {
            if (st.row == NO_ROW) {
                run_gadget(p, p.gadgets[st.sub], s);
            } else {
                SlotRow r{s, p.row_slots.data() + p.row_off[st.row], st.row};
                const vpbs_gate& g = p.gates[p.row_gate[st.row]];
                gen_run(g, st.sub, p.consts.data() + (size_t)st.row * mc, r, tables_of(g));
            }
        } catch (const GenError& e) {
            err = e.what + (st.row == NO_ROW ? " (generator " + std::to_string(st.sub) + ")" : " (row " + std::to_string(st.row) + ")");
            return VPBS_ERR_INVALID;
        }
        if (!s.error.empty()) {
            err = s.error;
            return VPBS_ERR_INVALID;
        }
    }
    lap("generators");
    // full_witness: every wire takes its class's value (unset -> 0), position ranges in parallel
    if (threads == 0) threads = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    auto fill = [&](unsigned t) {
        const size_t lo = p.total * t / threads, hi = p.total * (t + 1) / threads;
        std::memset(wires_out + lo, 0, (hi - lo) * sizeof(u64));
        const size_t a = std::lower_bound(p.out_pos.begin(), p.out_pos.end(), (u32)lo) - p.out_pos.begin();
        for (size_t i = a; i < p.out_pos.size() && p.out_pos[i] < hi; ++i) wires_out[p.out_pos[i]] = s.val[p.out_slot[i]];
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(fill, t);
    fill(0);
    for (auto& th : pool) th.join();
    lap("full_witness");
    return VPBS_OK;
}
// ---- two-phase runs ----
// Taint propagation along the schedule: the late presets taint their slots, a step that reads a tainted slot is late and taints what it
// writes -- unless the early phase knows that value anyway (an early preset, or an early step writes it too: the late step is then the
// second writer of the slot and only compares; e.g. the limbs a late range check connects to the constant zero must not make every
// reader of zero late).  Reads / writes of a gate generator are found the way plan creation finds them (gen_run on flags).
unsigned late_phase_threads();   // further down: the pool size of the late phase (environment, host setting, CPU budget of the process)

int plan_split(vpbs_witness_plan* pp, const uint8_t* late, std::string& err) {
    if (!pp || !late) {
        err = "malformed arguments";
        return VPBS_ERR_INVALID;
    }
    vpbs_witness_plan& p = *pp;
    // taint[slot]: 0 = the early phase has the value; k > 0 = it exists once late stage k has run (late[i] is the stage preset i arrives
    // in: 1, 2, ...; a step belongs to the highest stage among what it reads)
    std::vector<uint8_t> taint(p.n_slots, 0), ready(p.n_slots, 1), known_early(p.n_slots, 0);
    p.preset_late.assign(late, late + p.preset_slot.size());
    unsigned n_stages = 1;
    for (size_t i = 0; i < p.preset_slot.size(); ++i) {
        if (!late[i]) known_early[p.preset_slot[i]] = 1;
        n_stages = std::max<unsigned>(n_stages, late[i]);
    }
    if (n_stages > 16) {
        err = "more than 16 late stages";
        return VPBS_ERR_INVALID;
    }
    for (size_t i = 0; i < p.preset_slot.size(); ++i)
        if (late[i] && !known_early[p.preset_slot[i]]) {
            uint8_t& t = taint[p.preset_slot[i]];
            t = t ? std::min(t, late[i]) : late[i];   // a class preset in two stages has its value from the first one on (the second compares)
        }
    auto wrote = [&](u32 slot, uint8_t stage) {
        if (!stage) {
            known_early[slot] = 1;
            taint[slot] = 0;   // a late step may have been the first to reach it in the schedule: the early phase has the value all the same
        } else if (!known_early[slot] && !taint[slot]) {
            taint[slot] = stage;   // the first writer in schedule order decides from when on readers may count on the value
        }
    };
    p.n_stages = n_stages;
    p.stage_presets.assign(n_stages, {});
    for (size_t i = 0; i < p.preset_slot.size(); ++i)
        if (late[i]) p.stage_presets[late[i] - 1].push_back((u32)i);
    p.step_late.assign(p.schedule.size(), 0);
    p.step_stage.assign(p.schedule.size(), 0);
    const unsigned mc = std::max(1u, p.max_consts);
    std::vector<unsigned> deps;
    std::vector<u32> written, written_w;
    // level bookkeeping, per phase: slot -> level of its (first) writer in that phase, 0 = a preset or a value of the other phase
    std::vector<u32> slot_level[2] = {std::vector<u32>(p.n_slots, 0), std::vector<u32>(p.n_slots, 0)};
    std::vector<u32> step_level(p.schedule.size(), 0);
    u32 max_level[2] = {0, 0};
    // what every step reads and writes (slots): [step_io_off[i], + step_n_in[i]) inputs, then outputs up to step_io_off[i + 1]
    std::vector<u32> step_io, step_io_off(p.schedule.size() + 1, 0), step_n_in(p.schedule.size(), 0);
    for (size_t i = 0; i < p.schedule.size(); ++i) {
        const auto& st = p.schedule[i];
        bool is_late = false;
        uint8_t stage = 0;
        u32 lvl = 1;
        if (st.row == NO_ROW) {
            const auto& gg = p.gadgets[st.sub];
            const u32* gs = p.gadget_slots.data() + gg.at;
            for (unsigned k = 0; k < gg.n_in; ++k) stage = std::max(stage, taint[gs[k]]);
            is_late = stage != 0;
            step_io.insert(step_io.end(), gs, gs + gg.n_in + gg.n_out);
            step_n_in[i] = gg.n_in;
            std::vector<u32>& sl = slot_level[is_late];
            for (unsigned k = 0; k < gg.n_in + gg.n_out; ++k) lvl = std::max(lvl, sl[gs[k]] + 1);
            for (unsigned k = 0; k < gg.n_out; ++k) {
                wrote(gs[gg.n_in + k], stage);
                // a late step that writes a value the early phase has is a comparer: nothing in the late phase waits for it
                if (sl[gs[gg.n_in + k]] == 0 && !(is_late && known_early[gs[gg.n_in + k]])) sl[gs[gg.n_in + k]] = lvl;
            }
        } else {
            const vpbs_gate& g = p.gates[p.row_gate[st.row]];
            const u32* rs = p.row_slots.data() + p.row_off[st.row];
            gen_deps(g, st.sub, deps);
            for (unsigned w : deps) stage = std::max(stage, taint[rs[w]]);
            is_late = stage != 0;
            written.clear();
            written_w.clear();
            FlagRow fr{ready, rs, written, written_w};
            try {
                gen_run(g, st.sub, p.consts.data() + (size_t)st.row * mc, fr, tables_of(g));
            } catch (const GenError& e) {
                err = e.what;
                return VPBS_ERR_INVALID;
            }
            for (unsigned w : deps) step_io.push_back(rs[w]);
            step_io.insert(step_io.end(), written.begin(), written.end());
            step_n_in[i] = (u32)deps.size();
            std::vector<u32>& sl = slot_level[is_late];
            for (unsigned w : deps) lvl = std::max(lvl, sl[rs[w]] + 1);
            for (u32 slot : written) lvl = std::max(lvl, sl[slot] + 1);   // an earlier writer of the phase: this one compares, after it
            for (u32 slot : written) {
                wrote(slot, stage);
                if (sl[slot] == 0 && !(is_late && known_early[slot])) sl[slot] = lvl;
            }
        }
        step_level[i] = lvl;
        max_level[is_late] = std::max(max_level[is_late], lvl);
        p.step_late[i] = is_late ? 1 : 0;
        p.step_stage[i] = stage;
        step_io_off[i + 1] = (u32)step_io.size();
    }
    // ---- chain lanes (see vpbs_witness_plan::Lane) ----
    constexpr u32 NONE = 0xFFFFFFFFu, MAX_LANES = 4;
    const u32 LANE_MIN_ROWS = std::getenv("VPBS_LANE_MIN_ROWS") ? (u32)std::atoi(std::getenv("VPBS_LANE_MIN_ROWS")) : 150;   // 0x7fffffff: no lanes
    std::vector<int> lane_of(p.schedule.size(), -1);
    for (int ph = 0; ph < 2; ++ph) {
        p.lanes[ph].clear();
        p.lane_steps_sorted[ph].clear();
        p.lane_pool[ph].reset();
        auto ins = [&](size_t i) { return std::make_pair(step_io.data() + step_io_off[i], step_io.data() + step_io_off[i] + step_n_in[i]); };
        auto outs = [&](size_t i) { return std::make_pair(step_io.data() + step_io_off[i] + step_n_in[i], step_io.data() + step_io_off[i + 1]); };
        // the first writer of every slot inside the phase; candidates = PoseidonGate rows that are the only writers of what they write
        std::vector<u32> writer(p.n_slots, NONE);
        std::vector<uint8_t> cand(p.schedule.size(), 0);
        for (size_t i = 0; i < p.schedule.size(); ++i) {
            if (p.step_late[i] != ph) continue;
            const auto& st = p.schedule[i];
            cand[i] = st.row != NO_ROW && p.gates[p.row_gate[st.row]].kind == VPBS_GATE_POSEIDON;
            for (auto o = outs(i); o.first != o.second; ++o.first) {
                if (writer[*o.first] == NONE) {
                    writer[*o.first] = (u32)i;
                } else {   // a second writer compares with the first: both stay in the levels, where that order is kept
                    cand[i] = 0;
                    cand[writer[*o.first]] = 0;
                }
            }
        }
        // a candidate whose result anything else of the phase reads is not one (and then neither are the rows it reads from): fixpoint
        std::vector<u32> work;
        for (size_t i = 0; i < p.schedule.size(); ++i)
            if (p.step_late[i] == ph && !cand[i]) work.push_back((u32)i);
        while (!work.empty()) {
            const u32 i = work.back();
            work.pop_back();
            for (auto d = ins(i); d.first != d.second; ++d.first) {
                const u32 w = writer[*d.first];
                if (w != NONE && w != i && cand[w]) {
                    cand[w] = 0;
                    work.push_back(w);
                }
            }
        }
        // chains of what is left: a row continues the chain of the row that produced its incoming capacity (wire 8 -- the sponge state a
        // hash carries from one permutation to the next); a row that starts from constants starts a chain.  Rows that merely READ another
        // chain's digest (the public-input hash absorbs the key hash) stay in their own chain and wait for it.
        std::vector<u32> chain_of(p.schedule.size(), NONE), chain_rows;
        size_t n_cand = 0;
        for (size_t i = 0; i < p.schedule.size(); ++i) {
            if (!cand[i]) continue;
            ++n_cand;
            const u32 w = step_n_in[i] > 8 ? writer[step_io[step_io_off[i] + 8]] : NONE;
            if (w != NONE && w != i && cand[w] && chain_of[w] != NONE) {
                chain_of[i] = chain_of[w];
            } else {
                chain_of[i] = (u32)chain_rows.size();
                chain_rows.push_back(0);
            }
            ++chain_rows[chain_of[i]];
        }
        if (n_cand < LANE_MIN_ROWS) continue;
        if (ph == 1 && n_stages > 1) continue;   // a staged late phase runs stage by stage: its (few) chain rows stay in the levels
        std::vector<std::pair<u32, u32>> comps;   // (rows, chain)
        for (u32 c = 0; c < chain_rows.size(); ++c) comps.push_back({chain_rows[c], c});
        std::sort(comps.begin(), comps.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        const size_t n_lanes = std::min<size_t>(MAX_LANES, comps.size());
        std::vector<u32> load(n_lanes, 0), lane_of_chain(chain_rows.size(), 0);
        for (const auto& c : comps) {   // longest chain first, each to the lane with the least work so far
            const size_t l = std::min_element(load.begin(), load.end()) - load.begin();
            load[l] += c.first;
            lane_of_chain[c.second] = (u32)l;
        }
        p.lanes[ph].resize(n_lanes);
        for (size_t i = 0; i < p.schedule.size(); ++i) {
            if (!cand[i]) continue;
            lane_of[i] = (int)lane_of_chain[chain_of[i]];
            p.lane_steps_sorted[ph].push_back((u32)i);
        }
        for (u32 i : p.lane_steps_sorted[ph]) {
            vpbs_witness_plan::Lane& L = p.lanes[ph][lane_of[i]];
            if (L.wait_off.empty()) L.wait_off.push_back(0);
            L.steps.push_back(i);
            for (auto d = ins(i); d.first != d.second; ++d.first) {
                const u32 w = writer[*d.first];
                if (w != NONE && lane_of[w] != lane_of[i]) L.wait.push_back(*d.first);   // produced by a level or by another lane
            }
            L.wait_off.push_back((u32)L.wait.size());
        }
    }
    p.late_stage.assign(n_stages > 1 ? n_stages : 0, vpbs_witness_plan::Phase{});
    for (unsigned q = 0; q < 2 + p.late_stage.size(); ++q) {
        // q = 0, 1: the early and the late phase as a whole; q = 2 + k: stage k + 1 of a staged late phase alone
        const int ph = q < 2 ? (int)q : 1;
        const uint8_t only_stage = q < 2 ? 0 : (uint8_t)(q - 1);
        vpbs_witness_plan::Phase& P = q < 2 ? p.phase[q] : p.late_stage[q - 2];
        auto mine = [&](size_t i) { return p.step_late[i] == ph && lane_of[i] < 0 && (!only_stage || p.step_stage[i] == only_stage); };
        // counting sort of the phase's steps by level (schedule order kept inside a level)
        P.level_off.assign(max_level[ph] + 2, 0);
        for (size_t i = 0; i < p.schedule.size(); ++i)
            if (mine(i)) ++P.level_off[step_level[i] + 1];
        for (u32 l = 1; l <= max_level[ph] + 1; ++l) P.level_off[l] += P.level_off[l - 1];
        P.order.assign(P.level_off[max_level[ph] + 1], 0);
        {
            std::vector<u32> at(P.level_off.begin(), P.level_off.end() - 1);
            for (size_t i = 0; i < p.schedule.size(); ++i)
                if (mine(i)) P.order[at[step_level[i]]++] = (u32)i;
        }
        if (only_stage) {   // a stage only walks the levels it has generators on
            std::vector<u32> off{0};
            for (u32 l = 0; l + 1 < P.level_off.size(); ++l)
                if (P.level_off[l + 1] != P.level_off[l]) off.push_back(P.level_off[l + 1]);
            P.level_off.swap(off);
        }
        // Estimated cost per generator (measured, VPBS_TRACE_WITNESS with one thread, EPYC 9575F): a PoseidonGate row is a whole
        // permutation (1.4 us), most others a handful of field operations (30 ns).  The threads of a wide level take equal COST, not
        // equal counts.
        P.cost.assign(P.order.size() + 1, 0);
        for (size_t k = 0; k < P.order.size(); ++k) {
            const auto& st = p.schedule[P.order[k]];
            u32 cost = 3;
            if (st.row == NO_ROW) {
                if (p.gadgets[st.sub].kind == VPBS_GEN_QUOTIENT_EXT) cost = 40;
            } else {
                switch (p.gates[p.row_gate[st.row]].kind) {
                    case VPBS_GATE_POSEIDON: cost = 140; break;
                    case VPBS_GATE_POSEIDON_MDS:
                    case VPBS_GATE_REDUCING:
                    case VPBS_GATE_REDUCING_EXT:
                    case VPBS_GATE_COSET_INTERPOLATION:
                    case VPBS_GATE_EXPONENTIATION: cost = 60; break;
                    case VPBS_GATE_BASE_SUM: cost = 20; break;
                    default: break;
                }
            }
            P.cost[k + 1] = P.cost[k] + cost;
        }
        if (q < 2) p.pool[ph].reset();
        // ---- strands of a late stage (vpbs_witness_plan::Phase) ----
        P.strand_threads = 0;
        P.strand_steps.clear(); P.strand_level.clear(); P.strand_off.clear();
        static const bool no_strands = std::getenv("VPBS_LATE_STRANDS") && std::atoi(std::getenv("VPBS_LATE_STRANDS")) == 0;
        if (only_stage && !no_strands && P.order.size() >= 512) {
            const unsigned T = late_phase_threads();
            // components: two generators of the stage belong together when one reads what the other writes during the stage, or both write
            // the same class during the stage (the second compares).  A value that exists before the stage starts -- early, a preset of a
            // stage <= this one, the result of an earlier stage -- ties nothing together.
            std::vector<uint8_t> preset_before(p.n_slots, 0);
            for (size_t i = 0; i < p.preset_slot.size(); ++i)
                if (late[i] && late[i] <= only_stage) preset_before[p.preset_slot[i]] = 1;
            auto known_before = [&](u32 slot) { return known_early[slot] || preset_before[slot] || (taint[slot] && taint[slot] < only_stage); };
            std::vector<u32> parent(P.order.size()), stage_writer(p.n_slots, NONE);
            for (u32 k = 0; k < parent.size(); ++k) parent[k] = k;
            auto find = [&](u32 x) {
                while (parent[x] != x) x = parent[x] = parent[parent[x]];
                return x;
            };
            auto unite = [&](u32 a, u32 b) {
                a = find(a); b = find(b);
                if (a != b) parent[std::max(a, b)] = std::min(a, b);
            };
            // in SCHEDULE order (a reader comes after a writer of each of its inputs); P.order is sorted by level, so map back
            std::vector<std::pair<u32, u32>> by_schedule(P.order.size());   // (schedule index, position in P.order)
            for (u32 k = 0; k < P.order.size(); ++k) by_schedule[k] = {P.order[k], k};
            std::sort(by_schedule.begin(), by_schedule.end());
            for (const auto& e : by_schedule) {
                const size_t i = e.first;
                for (u32 x = step_io_off[i]; x < step_io_off[i] + step_n_in[i]; ++x)
                    if (stage_writer[step_io[x]] != NONE) unite(e.second, stage_writer[step_io[x]]);
                for (u32 x = step_io_off[i] + step_n_in[i]; x < step_io_off[i + 1]; ++x) {
                    const u32 slot = step_io[x];
                    if (known_before(slot)) continue;
                    if (stage_writer[slot] == NONE) stage_writer[slot] = e.second;
                    else unite(e.second, stage_writer[slot]);
                }
            }
            std::vector<u64> comp_cost(P.order.size(), 0);
            for (u32 k = 0; k < P.order.size(); ++k) comp_cost[find(k)] += P.cost[k + 1] - P.cost[k];
            std::vector<std::pair<u64, u32>> comps;
            for (u32 k = 0; k < P.order.size(); ++k)
                if (comp_cost[k]) comps.push_back({comp_cost[k], k});
            std::sort(comps.begin(), comps.end(), [](const auto& a, const auto& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
            std::vector<u64> load(T, 0);
            std::vector<u32> bin_of(P.order.size(), 0);
            for (const auto& c : comps) {   // heaviest first, each to the lightest bin
                const u32 b = (u32)(std::min_element(load.begin(), load.end()) - load.begin());
                load[b] += c.first;
                bin_of[c.second] = b;
            }
            const u64 total = P.cost[P.order.size()], heaviest = *std::max_element(load.begin(), load.end());
            if (T >= 2 && heaviest * T <= total + total / 2) {   // balanced within 1.5 x: worth it (one giant component is not)
                std::vector<u32> cnt(T + 1, 0);
                for (u32 k = 0; k < P.order.size(); ++k) ++cnt[bin_of[find(k)] + 1];
                for (u32 b = 0; b < T; ++b) cnt[b + 1] += cnt[b];
                P.strand_off = cnt;
                P.strand_steps.assign(P.order.size(), 0);
                P.strand_level.assign(P.order.size(), 0);
                std::vector<u32> at(cnt.begin(), cnt.end() - 1);
                for (u32 k = 0; k < P.order.size(); ++k) {   // P.order is sorted by level: so is every strand
                    const u32 b = bin_of[find(k)];
                    P.strand_steps[at[b]] = P.order[k];
                    P.strand_level[at[b]++] = step_level[P.order[k]];
                }
                P.strand_threads = T;
            }
            if (std::getenv("VPBS_TRACE_WITNESS"))
                std::fprintf(stderr, "[witness split] late stage %u: %zu components, heaviest %.1f %% of the stage, %u bins, heaviest bin %.2f x the mean -> %s\n",
                             (unsigned)only_stage, comps.size(), comps.empty() ? 0.0 : 100.0 * comps[0].first / std::max<u64>(1, total), T,
                             total ? (double)heaviest * T / total : 0.0, P.strand_threads ? "strands" : "levels");
        }
    }
    // the late wire positions, ordered by the stage their value exists from and then by value slot: a stage's share is one contiguous range
    // of the packed output (written as soon as the stage has run), and packing reads the slot array front to back
    p.late_out.clear();
    for (size_t i = 0; i < p.out_slot.size(); ++i)
        if (taint[p.out_slot[i]]) p.late_out.push_back((u32)i);
    std::stable_sort(p.late_out.begin(), p.late_out.end(), [&](u32 a, u32 b) {
        const u32 sa = p.out_slot[a], sb = p.out_slot[b];
        return taint[sa] != taint[sb] ? taint[sa] < taint[sb] : sa < sb;
    });
    p.late_out_stage_off.assign(n_stages + 1, 0);
    for (u32 i : p.late_out) ++p.late_out_stage_off[taint[p.out_slot[i]]];
    for (unsigned k = 1; k <= n_stages; ++k) p.late_out_stage_off[k] += p.late_out_stage_off[k - 1];
    // the early-known values the late phase touches: what its steps read, what they write as comparers, what late presets are compared with
    {
        std::vector<uint8_t> seen(p.n_slots, 0);
        p.late_in_slots.clear();
        auto touch = [&](u32 slot) {
            if (!taint[slot] && !seen[slot]) {
                seen[slot] = 1;
                p.late_in_slots.push_back(slot);
            }
        };
        for (size_t i = 0; i < p.preset_slot.size(); ++i)
            if (late[i]) touch(p.preset_slot[i]);
        for (size_t i = 0; i < p.schedule.size(); ++i)
            if (p.step_late[i])
                for (u32 k = step_io_off[i]; k < step_io_off[i + 1]; ++k) touch(step_io[k]);
        std::sort(p.late_in_slots.begin(), p.late_in_slots.end());
    }
    p.pack_runs.clear();
    for (unsigned st = 0; st < n_stages; ++st)
        for (size_t k = p.late_out_stage_off[st]; k < p.late_out_stage_off[st + 1]; ++k) {
            const u32 sl = p.out_slot[p.late_out[k]];
            if (k > p.late_out_stage_off[st] && p.pack_runs.back().slot + p.pack_runs.back().len == sl) ++p.pack_runs.back().len;
            else p.pack_runs.push_back({(u32)k, sl, 1u});
        }
    p.late_slot_runs.clear();
    for (size_t sl = 0; sl < p.n_slots; ++sl) {
        if (!taint[sl]) continue;
        if (!p.late_slot_runs.empty() && p.late_slot_runs.back().first + p.late_slot_runs.back().second == sl) ++p.late_slot_runs.back().second;
        else p.late_slot_runs.push_back({(u32)sl, 1u});
    }
    // the early phase alone as a device schedule (vpbs_witness_device_create_early)
    build_device_schedule(p, p.dev_early, p.step_out, p.step_out_w, p.step_out_off, p.step_late.data(), p.preset_late.data());
    {   // ... and the late phase alone, on top of the early phase's values
        std::vector<uint8_t> early_step(p.schedule.size()), early_preset(p.preset_slot.size()), early_known(p.n_slots);
        for (size_t i = 0; i < early_step.size(); ++i) early_step[i] = !p.step_late[i];
        for (size_t i = 0; i < early_preset.size(); ++i) early_preset[i] = !p.preset_late[i];
        for (size_t sl = 0; sl < p.n_slots; ++sl) early_known[sl] = !taint[sl];
        build_device_schedule(p, p.dev_late, p.step_out, p.step_out_w, p.step_out_off, early_step.data(), early_preset.data(), early_known.data());
        // ... and stage by stage: stage k over what the early phase and the stages before it have set (the slots the stages REALLY write,
        // in stage order -- a class two stages write is written by the first to run and compared by the other, whatever the schedule order)
        p.dev_late_stage.clear();
        p.late_row_slots.clear();
        if (n_stages > 1 && p.dev_late.supported) {
            std::vector<uint8_t> known(early_known), skip_step(p.schedule.size()), skip_preset(p.preset_slot.size());
            p.dev_late_stage.assign(n_stages, {});
            p.late_row_slots = p.row_slots;
            for (unsigned k = 1; k <= n_stages; ++k) {
                for (size_t i = 0; i < skip_step.size(); ++i) skip_step[i] = !(p.step_late[i] && p.step_stage[i] == k);
                for (size_t i = 0; i < skip_preset.size(); ++i) skip_preset[i] = p.preset_late[i] != k;
                auto& ds = p.dev_late_stage[k - 1];
                build_device_schedule(p, ds, p.step_out, p.step_out_w, p.step_out_off, skip_step.data(), skip_preset.data(), known.data());
                if (!ds.supported) {
                    p.dev_late_stage.clear();
                    p.late_row_slots.clear();
                    break;
                }
                for (size_t j = 0; j < ds.row_slots.size(); ++j)
                    if (ds.row_slots[j] != p.row_slots[j]) p.late_row_slots[j] = ds.row_slots[j];
                std::vector<u32>().swap(ds.row_slots);   // the shared table stands for it
                for (size_t i = 0; i < skip_preset.size(); ++i)
                    if (!skip_preset[i]) known[p.preset_slot[i]] = 1;
                for (size_t i = 0; i < skip_step.size(); ++i)
                    if (!skip_step[i])
                        for (u32 o = p.step_out_off[i]; o < p.step_out_off[i + 1]; ++o) known[p.step_out[o]] = 1;
            }
        }
    }
    if (std::getenv("VPBS_TRACE_WITNESS")) {
        std::vector<uint8_t> row_late(p.n, 0);
        for (u32 i : p.late_out) row_late[p.out_pos[i] % p.n] = 1;
        size_t rows = 0, lo = p.n, hi = 0, runs = 0;
        for (size_t r = 0; r < p.n; ++r)
            if (row_late[r]) {
                ++rows;
                lo = std::min(lo, r);
                hi = r;
                if (r == 0 || !row_late[r - 1]) ++runs;
            }
        std::fprintf(stderr, "[witness split] %zu late wire positions on %zu rows in [%zu, %zu], %zu runs of consecutive rows\n", p.late_out.size(), rows,
                     lo, hi, runs);
        for (size_t k = 0; k < p.late_stage.size(); ++k) {
            const vpbs_witness_plan::Phase& P = p.late_stage[k];
            size_t pos_rows = 0, wide = 0;
            for (u32 i : P.order)
                if (p.schedule[i].row != NO_ROW && p.gates[p.row_gate[p.schedule[i].row]].kind == VPBS_GATE_POSEIDON) ++pos_rows;
            for (size_t l = 0; l + 1 < P.level_off.size(); ++l)
                if (P.cost[P.level_off[l + 1]] - P.cost[P.level_off[l]] >= 600) ++wide;
            std::fprintf(stderr, "[witness split] late stage %zu: %zu generators (%zu PoseidonGate rows) on %zu levels, %zu of them wide\n", k + 1,
                         P.order.size(), pos_rows, P.level_off.size() - 1, wide);
        }
    }
    if (std::getenv("VPBS_TRACE_WITNESS") && p.dev_late.supported) {
        // the late phase as the DEVICE walks it (wd_walk_kernel: one level at a time, a level costs its slowest share): PoseidonGate rows per level
        const auto& ds = p.dev_late;
        size_t rows = 0, rowops = 0, gt32 = 0, gt64 = 0, gt128 = 0, passes512 = 0, passes1024 = 0, passes4096 = 0, with_rowops = 0;
        for (u32 l = 1; l <= ds.n_levels; ++l) {
            const size_t c = ds.poseidon_off[l + 1] - ds.poseidon_off[l], ro = ds.rowops_off[l + 1] - ds.rowops_off[l];
            rows += c;
            rowops += ro;
            with_rowops += ro != 0;
            gt32 += c > 32;
            gt64 += c > 64;
            gt128 += c > 128;
            passes512 += (c + 31) / 32;
            passes1024 += (c + 63) / 64;
            passes4096 += (c + 255) / 256;
        }
        for (size_t k = 0; k < p.dev_late_stage.size(); ++k) {
            const auto& sd = p.dev_late_stage[k];
            size_t widest = 0;
            for (u32 l = 1; l <= sd.n_levels; ++l)
                widest = std::max<size_t>(widest, (sd.poseidon_off[l + 1] - sd.poseidon_off[l]) + (sd.arith_off[l + 1] - sd.arith_off[l]) +
                                                      (sd.rowops_off[l + 1] - sd.rowops_off[l]) + (sd.misc_off[l + 1] - sd.misc_off[l]) + (sd.bits_off[l + 1] - sd.bits_off[l]));
            std::fprintf(stderr, "[witness split] device late stage %zu: %u levels, %zu PoseidonGate rows, %zu row operations, %zu arithmetic, %zu bits, %zu misc, "
                         "%zu constants; widest level %zu operations\n", k + 1, sd.n_levels, sd.poseidon.size(), sd.rowops.size(), sd.arith.size(), sd.bits.size(),
                         sd.misc.size(), sd.consts.size(), widest);
        }
        {   // the row operations (one thread each in the walk) by gate kind: how many, on how many levels
            std::map<unsigned, std::pair<size_t, size_t>> by_kind;
            for (u32 l = 1; l <= ds.n_levels; ++l) {
                std::map<unsigned, size_t> here;
                for (u32 i = ds.rowops_off[l]; i < ds.rowops_off[l + 1]; ++i) ++here[p.gates[p.row_gate[ds.rowops[i].row]].kind];
                for (auto& kv : here) {
                    by_kind[kv.first].first += kv.second;
                    ++by_kind[kv.first].second;
                }
            }
            for (auto& kv : by_kind)
                std::fprintf(stderr, "[witness split] device late schedule: gate kind %u: %zu row operations on %zu levels\n", kv.first, kv.second.first, kv.second.second);
        }
        std::fprintf(stderr, "[witness split] device late schedule: %u levels, %zu PoseidonGate rows (levels with > 32 / 64 / 128 rows: %zu / %zu / %zu; "
                     "row passes with 512 / 1024 / 4096 threads: %zu / %zu / %zu), %zu row operations on %zu levels, %zu arithmetic, %zu bits, %zu misc\n",
                     ds.n_levels, rows, gt32, gt64, gt128, passes512, passes1024, passes4096, rowops, with_rowops, ds.arith.size(), ds.bits.size(), ds.misc.size());
    }
    p.is_split = true;
    return VPBS_OK;
}
}  // namespace
}  // namespace vpbs

namespace vpbs {
namespace {
// touches the pages of a fresh state under the slots only the late phase writes (nothing reads a value whose flag is clear, so the zero
// written here is never seen); values the state already holds are left alone
void prefault_late_slots(const vpbs_witness_plan& p, SlotState& s) {
    constexpr uintptr_t PAGE = 4096;
    uintptr_t last = 0;
    for (const auto& run : p.late_slot_runs) {
        for (u32 k = 0; k < run.second;) {
            const u32 slot = run.first + k;
            const uintptr_t page = reinterpret_cast<uintptr_t>(&s.val[slot]) / PAGE;
            if (page != last && !s.is_set[slot]) {
                s.val[slot] = 0;
                last = page;
            }
            const uintptr_t next_page_addr = (page + 1) * PAGE;
            const u32 skip = (u32)((next_page_addr - reinterpret_cast<uintptr_t>(&s.val[slot]) + sizeof(u64) - 1) / sizeof(u64));
            k += std::max(1u, skip);
        }
    }
}
}  // namespace
}  // namespace vpbs

namespace vpbs {
namespace {
// packed_out[k] = value of the slot behind late_out[k] (0 where nothing set it), k in [lo, hi): range copies along pack_runs
void pack_late_range(const vpbs_witness_plan& p, const SlotState& s, u64* packed_out, size_t lo, size_t hi) {
    if (lo >= hi) return;
    auto it = std::upper_bound(p.pack_runs.begin(), p.pack_runs.end(), lo,
                               [](size_t k, const vpbs_witness_plan::PackRun& r) { return k < r.k; });
    --it;   // the run that holds lo (runs cover every k from 0)
    for (; it != p.pack_runs.end() && it->k < hi; ++it) {
        const size_t k0 = std::max<size_t>(it->k, lo), k1 = std::min<size_t>((size_t)it->k + it->len, hi);
        const u64* v = &s.val[it->slot + (k0 - it->k)];
        const uint8_t* f = &s.is_set[it->slot + (k0 - it->k)];
        u64* out = packed_out + k0;
        for (size_t i = 0, n = k1 - k0; i < n; ++i) out[i] = f[i] ? v[i] : 0;
    }
}
}  // namespace
}  // namespace vpbs

struct vpbs_witness_state {
    vpbs::SlotState s;
    unsigned stages_done = 0;   // late stages that have run on this state (vpbs_witness_plan_run_late_stage)
    unsigned stages_packed = 0; // ... and how many of them have their wires in the caller's packed buffer already
    // the late pool was left spinning for the next stage of this state: the state holds a share of the pool and the token of that promise;
    // whoever runs the stage (or frees the state) ends the spinning -- unless another run took the pool meanwhile (LevelPool::end_if_still_kept)
    std::shared_ptr<vpbs::LevelPool> awake;
    unsigned long awake_token = 0;
    void release_awake() {
        if (awake) awake->end_if_still_kept(awake_token);
        awake.reset();
        awake_token = 0;
    }
    int failed = 0;             // a stage run ahead failed with this status (a generator rejected its inputs; not only slot mismatches are sticky):
    std::string failure;        // ... every later stage and run_late[_packed] report it again instead of building on a half-run stage
    ~vpbs_witness_state() { release_awake(); }
};

namespace vpbs {
namespace {
// one step of the schedule on the slot state; returns false (message in err) when a generator rejects its inputs
bool run_one(const vpbs_witness_plan& p, SlotState& s, size_t i, unsigned mc, std::string& err) {
    const auto& st = p.schedule[i];
    try {
        if (st.row == NO_ROW) {
            run_gadget(p, p.gadgets[st.sub], s);
        } else {
            SlotRow r{s, p.row_slots.data() + p.row_off[st.row], st.row};
            const vpbs_gate& g = p.gates[p.row_gate[st.row]];
            gen_run(g, st.sub, p.consts.data() + (size_t)st.row * mc, r, tables_of(g));
        }
    } catch (const GenError& e) {
        err = e.what + (st.row == NO_ROW ? " (generator " + std::to_string(st.sub) + ")" : " (row " + std::to_string(st.row) + ")");
        return false;
    }
    return true;
}

#if defined(VPBS_HAVE_POSEIDON_X8)
// Up to eight PoseidonGate rows of one dependency level at once (they are independent: plan_split), one permutation per AVX-512 lane:
// poseidon_generate for `cnt` schedule steps.  The wires written are the scalar generator's, bit for bit (canonical values).
__attribute__((target("avx512f,avx512dq")))
bool poseidon_rows_x8(const vpbs_witness_plan& p, SlotState& s, const u32* steps, unsigned cnt, std::string& err) {
    using poseidon_x8::V;
    alignas(64) u64 in[12][8];
    const u32* rs[8];
    u32 row[8];
    for (unsigned l = 0; l < 8; ++l) {
        const unsigned q = l < cnt ? l : 0;   // spare lanes repeat the first row
        const u32 r = p.schedule[steps[q]].row;
        row[l] = r;
        rs[l] = p.row_slots.data() + p.row_off[r];
        if (l >= cnt) {
            for (int i = 0; i < 12; ++i) in[i][l] = in[i][0];
            continue;
        }
        SlotRow sr{s, rs[l], r};
        const u64 swap = sr.get(24);
        if (swap > 1) {
            err = "PoseidonGate: swap wire is not boolean (row " + std::to_string(r) + ")";
            return false;
        }
        for (int i = 0; i < 4; ++i) {
            const u64 lhs = sr.get(i), rhs = sr.get(i + 4);
            const u64 delta = gl::mul(swap, gl::sub(rhs, lhs));
            sr.set(25 + i, delta);
            in[i][l] = gl::add(lhs, delta);
            in[i + 4][l] = gl::sub(rhs, delta);
        }
        for (int i = 8; i < 12; ++i) in[i][l] = sr.get(i);
    }
    V st[12], gate[106];
    for (int i = 0; i < 12; ++i) st[i] = _mm512_load_si512(in[i]);
    poseidon_x8::permute(st, gate);
    alignas(64) u64 out[8];
    auto scatter = [&](V v, unsigned wire) __attribute__((target("avx512f,avx512dq"))) {
        _mm512_store_si512(out, v);
        for (unsigned l = 0; l < cnt; ++l) s.set(rs[l][wire], out[l], (u32)(wire * s.n + row[l]));
    };
    // wires 29 .. 134 = gate[0 .. 105] (S-box inputs of full rounds 1-3, of the 22 partial rounds, of full rounds 26-29).  Nothing but the
    // wire matrix reads them, and on every row the plan found them on 106 consecutive slots of their own (poseidon_private_base) they go
    // there as one range per row -- the checked single stores cost more than the permutation (9 against 8 us per eight rows, measured)
    bool ranges = true;
    for (unsigned l = 0; l < cnt; ++l) ranges = ranges && p.poseidon_private_base[row[l]] != NONE && !s.is_set[p.poseidon_private_base[row[l]]];
    if (ranges) {
        alignas(64) u64 lanes[106][8];
        const V P = _mm512_set1_epi64((long long)gl::P);
        for (int j = 0; j < 106; ++j) _mm512_store_si512(lanes[j], _mm512_min_epu64(gate[j], _mm512_sub_epi64(gate[j], P)));   // canonical, as set() leaves them
        for (unsigned l = 0; l < cnt; ++l) {
            const u32 base = p.poseidon_private_base[row[l]];
            u64* dst = &s.val[base];
            for (int j = 0; j < 106; ++j) dst[j] = lanes[j][l];
            std::memset(&s.is_set[base], 1, 106);
        }
        __atomic_thread_fence(__ATOMIC_RELEASE);
    } else {
        for (int round = 1; round < 4; ++round)
            for (int i = 0; i < 12; ++i) scatter(gate[12 * (round - 1) + i], 29 + 12 * (round - 1) + i);
        for (int q = 0; q < 22; ++q) scatter(gate[36 + q], 65 + q);
        for (int round = 26; round < 30; ++round)
            for (int i = 0; i < 12; ++i) scatter(gate[58 + 12 * (round - 26) + i], 87 + 12 * (round - 26) + i);
    }
    for (int i = 0; i < 12; ++i) scatter(st[i], 12 + i);
    return true;
}
#endif

// VPBS_TRACE_WITNESS: where a sequential run spends its time, per gate kind (0..31) / gadget generator kind (32..)
struct KindProfile {
    const bool on = std::getenv("VPBS_TRACE_WITNESS") != nullptr;
    double us[64] = {0};
    unsigned count[64] = {0};
    std::chrono::steady_clock::time_point t;
    void start() {
        if (on) t = std::chrono::steady_clock::now();
    }
    void stop(const vpbs_witness_plan& p, size_t i) {
        if (!on) return;
        const auto& st = p.schedule[i];
        const unsigned kind = st.row == NO_ROW ? 32 + (p.gadgets[st.sub].kind & 31) : (p.gates[p.row_gate[st.row]].kind & 31);
        us[kind] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t).count();
        ++count[kind];
    }
    void report(const char* phase) const {
        if (!on) return;
        for (unsigned q = 0; q < 64; ++q)
            if (count[q])
                std::fprintf(stderr, "[witness %s] %s kind %2u: %6u generators, %8.1f us (%.2f us each)\n", phase, q < 32 ? "gate  " : "gadget", q & 31,
                             count[q], us[q], us[q] / count[q]);
    }
};

// The generators of one phase of a split plan, level by level.  A level worth at least PAR_MIN_COST is shared by the threads of the
// phase's pool, each taking an equal share of the level's estimated cost; cheaper levels (a single PoseidonGate row of a hash chain, a
// few field operations) stay on the calling thread.  Generators of one level touch disjoint slots (plan_split), so the slot state needs no
// locking; a "set twice with different values" report goes through SlotState's mutex.  `after` (may be empty) is one more shared job at
// the end -- the wires of the phase written into the matrix.
bool lanes_of_phase_empty(const vpbs_witness_plan& p, int ph) { return p.lane_steps_sorted[ph].empty(); }

// stage: 0 = the whole phase; k > 0 (late phase of a plan split into stages) = the generators of late stage k only.
// before (may be empty): a shared job in front of the generators (the presets of the stage); keep_awake: leave the pool's workers spinning
// at the end -- the next stage of this state follows within a fraction of a millisecond (whoever runs it, or frees the state, ends the pool).
int run_levels(const vpbs_witness_plan& p, int ph, unsigned stage, SlotState& s, unsigned threads,
               const std::function<void(unsigned, unsigned)>& after, std::string& err,
               const std::function<void(unsigned, unsigned)>& before = std::function<void(unsigned, unsigned)>(), bool keep_awake = false,
               vpbs_witness_state* awake_for = nullptr) {
    constexpr u32 PAR_MIN_COST = 600;   // ~6 us of work: four PoseidonGate rows
    const vpbs_witness_plan::Phase& P = stage ? p.late_stage[stage - 1] : p.phase[ph];
    const unsigned mc = std::max(1u, p.max_consts);
    const u32 n_levels = (u32)P.level_off.size() - 1;
    static const bool trace = std::getenv("VPBS_TRACE_WITNESS") != nullptr;
    const char* name = ph ? "late" : "early";
    LevelPool* pool = nullptr;
    std::shared_ptr<LevelPool> pool_share;
    std::unique_lock<std::mutex> busy;
    if (threads > 1) {
        {
            std::lock_guard<std::mutex> lk(p.pool_mutex);
            if (!p.pool[ph]) p.pool[ph].reset(new LevelPool(threads, ph ? "vpbs-late-pool" : "vpbs-early-pool"));   // the phase's pool keeps the size of its first run
            pool_share = p.pool[ph];
            pool = pool_share.get();
        }
        busy = std::unique_lock<std::mutex>(pool->busy, std::try_to_lock);
        if (!busy.owns_lock()) pool = nullptr;   // another run of this phase has the threads: this one goes alone
        else pool->taken_by_a_run();             // a promise an earlier run left behind is void: the workers are this run's
    }
    auto promise_awake = [&] {
        if (!awake_for) return;
        awake_for->awake = pool_share;
        awake_for->awake_token = pool->keep_awake_token();
    };
    if (!pool) {   // alone: the plan's own schedule order (generators of a row's neighbourhood together: far fewer cache misses than level order)
        if (before) before(0, 1);
        if (s.failed()) {
            err = s.error_text();
            return VPBS_ERR_INVALID;
        }
#if defined(VPBS_HAVE_POSEIDON_X8)
        // ... except for the LATE phase where the CPU has AVX-512: its cost is its PoseidonGate rows (3 649 of 31 613 generators of the cyclic
        // circuit's in-circuit verifier, 1.2 us each one by one), and the rows of one dependency level are independent -- level by level, eight
        // rows per permutation (0.41 us each): what a rank with ONE CPU for the late phase runs (4.5 -> 2 ms per chained step)
        if (ph == 1 && poseidon_x8::enabled() && !std::getenv("VPBS_TRACE_WITNESS")) {
            u32 pending[8];
            unsigned np = 0;
            bool ok = true;
            auto flush = [&] {
                if (np >= 3) ok = ok && poseidon_rows_x8(p, s, pending, np, err);
                else
                    for (unsigned q = 0; q < np; ++q) ok = ok && run_one(p, s, pending[q], mc, err);
                np = 0;
            };
            for (u32 l = 0; l < n_levels && ok; ++l) {
                for (u32 k = P.level_off[l]; k < P.level_off[l + 1] && ok; ++k) {
                    const auto& st = p.schedule[P.order[k]];
                    if (st.row != NO_ROW && p.gates[p.row_gate[st.row]].kind == VPBS_GATE_POSEIDON) {
                        pending[np++] = P.order[k];
                        if (np == 8) flush();
                    } else {
                        ok = run_one(p, s, P.order[k], mc, err);
                    }
                }
                flush();
            }
            if (ok && !lanes_of_phase_empty(p, ph))
                for (u32 i : p.lane_steps_sorted[ph])
                    if (!(ok = run_one(p, s, i, mc, err))) break;
            if (!ok || s.failed()) {
                if (err.empty()) err = s.error_text();
                return VPBS_ERR_INVALID;
            }
            if (after) after(0, 1);
            return VPBS_OK;
        }
#endif
        KindProfile prof;
        for (size_t i = 0; i < p.schedule.size(); ++i) {
            if (p.step_late[i] != ph || (stage && p.step_stage[i] != stage)) continue;
            prof.start();
            const bool ok = run_one(p, s, i, mc, err);
            prof.stop(p, i);
            if (!ok) return VPBS_ERR_INVALID;
            if (s.failed()) {
                err = s.error_text();
                return VPBS_ERR_INVALID;
            }
        }
        prof.report(name);
        if (after) after(0, 1);
        return VPBS_OK;
    }
    threads = pool->threads;
#if defined(VPBS_HAVE_POSEIDON_X8)
    const bool x8 = poseidon_x8::enabled();   // vpbs_host_set_poseidon_x8
#endif
    std::atomic<bool> failed{false};
    std::vector<std::string> errs(threads);
    if (stage && P.strand_threads == threads) {
        // STRANDS: every thread runs its bin of independent components from the first level to the last, no barrier in between
        const double t_begin = trace ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0;
        pool->begin();
        if (before) pool->share([&](unsigned t) { before(t, threads); });
        std::vector<double> t_start(threads, 0), t_stop(threads, 0);   // trace: when each thread picked its strand up, when it was through
        auto clock_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        pool->share([&](unsigned t) {
            if (trace) t_start[t] = clock_ms();
            struct Stop {
                double& at;
                bool on;
                ~Stop() {
                    if (on) at = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
                }
            } stop_clock{t_stop[t], trace};
            const u32 lo = P.strand_off[t], hi = P.strand_off[t + 1];
#if defined(VPBS_HAVE_POSEIDON_X8)
            u32 pending[8];
            unsigned np = 0;
            auto flush = [&] {
                if (np >= 3) {
                    if (!poseidon_rows_x8(p, s, pending, np, errs[t])) failed.store(true);
                } else {
                    for (unsigned q = 0; q < np; ++q)
                        if (!run_one(p, s, pending[q], mc, errs[t])) failed.store(true);
                }
                np = 0;
            };
#endif
            for (u32 k = lo; k < hi && !failed.load(std::memory_order_relaxed); ++k) {
                const u32 i = P.strand_steps[k];
#if defined(VPBS_HAVE_POSEIDON_X8)
                // the PoseidonGate rows of one level are independent: eight at a time; everything else of the level right away (a later
                // level may read either, so the batch is flushed when the level changes)
                if (k > lo && P.strand_level[k] != P.strand_level[k - 1]) flush();
                const auto& st = p.schedule[i];
                if (x8 && st.row != NO_ROW && p.gates[p.row_gate[st.row]].kind == VPBS_GATE_POSEIDON) {
                    pending[np++] = i;
                    if (np == 8) flush();
                    continue;
                }
#endif
                if (!run_one(p, s, i, mc, errs[t])) failed.store(true);
            }
#if defined(VPBS_HAVE_POSEIDON_X8)
            if (!failed.load(std::memory_order_relaxed)) flush();
#endif
        });
        int rc = failed.load() || s.failed() ? VPBS_ERR_INVALID : VPBS_OK;
        const double t_mid = trace ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() : 0;
        if (rc == VPBS_OK && after) pool->share([&](unsigned t) { after(t, threads); });
        if (keep_awake && rc == VPBS_OK && awake_for) {
            promise_awake();
        } else {
            pool->end();
        }
        if (trace) {
            double late_start = 0, longest = 0, shortest = 1e9;
            for (unsigned t = 0; t < threads; ++t) {
                late_start = std::max(late_start, t_start[t] - t_begin);
                longest = std::max(longest, t_stop[t] - t_start[t]);
                shortest = std::min(shortest, t_stop[t] - t_start[t]);
            }
            std::fprintf(stderr, "[witness %s] stage %u, %u threads: %zu generators in strands %.2f ms (last thread started after %.2f ms; strands took "
                         "%.2f .. %.2f ms), wires %.2f ms\n", name, stage, threads, P.order.size(), t_mid - t_begin, late_start, shortest, longest,
                         clock_ms() - t_mid);
        }
        if (rc != VPBS_OK) {
            for (const auto& e : errs)
                if (err.empty() && !e.empty()) err = e;
            if (err.empty()) {
                const std::string set_twice = s.error_text();
                err = set_twice.empty() ? "a generator failed" : set_twice;
            }
        }
        return rc;
    }
    // the presets of the phase / stage first: the chain lanes started below read them without waiting (they only wait for what levels and
    // other lanes produce) -- ThreadSanitizer found the lanes starting ahead of the presets when this job moved into the pool
    bool pool_awake = false;
    if (before) {
        if (P.order.size() >= 64) {   // worth waking the pool for (a stage of a few generators sets its handful of presets here)
            pool->begin();
            pool_awake = true;
            pool->share([&](unsigned t) { before(t, threads); });
        } else {
            before(0, 1);
        }
    }
    // the chain lanes start now, on threads of their own, and run next to the levels
    const std::vector<vpbs_witness_plan::Lane>& lanes = p.lanes[ph];
    LevelPool* lane_pool = nullptr;
    std::unique_lock<std::mutex> lane_busy;
    std::vector<std::string> lane_errs(lanes.size() + 1);
    if (!lanes.empty()) {
        {
            std::lock_guard<std::mutex> lk(p.pool_mutex);
            if (!p.lane_pool[ph]) p.lane_pool[ph].reset(new LevelPool((unsigned)lanes.size() + 1, ph ? "vpbs-late-lane" : "vpbs-early-lane"));
            lane_pool = p.lane_pool[ph].get();
        }
        lane_busy = std::unique_lock<std::mutex>(lane_pool->busy, std::try_to_lock);
        if (!lane_busy.owns_lock()) lane_pool = nullptr;
    }
    auto run_lane = [&](unsigned t) {
        const vpbs_witness_plan::Lane& L = lanes[t - 1];
        for (size_t k = 0; k < L.steps.size(); ++k) {
            for (u32 w = L.wait_off[k]; w < L.wait_off[k + 1]; ++w)
                spin_until([&] { return s.ready(L.wait[w]) || failed.load(std::memory_order_relaxed); });
            if (failed.load(std::memory_order_relaxed)) return;
            if (!run_one(p, s, L.steps[k], mc, lane_errs[t])) {
                failed.store(true);
                return;
            }
        }
    };
    if (lane_pool) {
        lane_pool->begin();
        lane_pool->start(run_lane);
    }
    double t_narrow = 0, t_wide = 0;
    u32 n_wide = 0;
    auto clock = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    // the workers are only needed up to the last wide level (and again for the wires at the end): in between they sleep
    u32 last_wide = 0;
    for (u32 l = 0; l < n_levels; ++l)
        if (P.cost[P.level_off[l + 1]] - P.cost[P.level_off[l]] >= PAR_MIN_COST) last_wide = l + 1;
    if (last_wide > 0 && !pool_awake) {
        pool->begin();
        pool_awake = true;
    }
    int rc = before && s.failed() ? VPBS_ERR_INVALID : VPBS_OK;
    for (u32 l = 0; l < n_levels && rc == VPBS_OK; ++l) {
        if (pool_awake && l >= last_wide) {
            pool->end();
            pool_awake = false;
        }
        const u32 lo = P.level_off[l], hi = P.level_off[l + 1];
        const double t_level = trace ? clock() : 0;
        if (P.cost[hi] - P.cost[lo] < PAR_MIN_COST) {
            for (u32 k = lo; k < hi && rc == VPBS_OK; ++k)
                if (!run_one(p, s, P.order[k], mc, errs[0])) rc = VPBS_ERR_INVALID;
            if (trace) t_narrow += clock() - t_level;
        } else {
            // The level is handed out in pieces of about equal estimated COST -- a piece is at least eight PoseidonGate rows' worth, and there
            // are about three pieces per thread -- claimed from a shared cursor: a thread the host has descheduled (the machine is shared)
            // holds up one piece, not an eighth of the level.
            const u32 *c0 = P.cost.data() + lo, *c1 = P.cost.data() + hi;
            const u64 total = *c1 - *c0;
            static const unsigned pieces_per_thread = std::getenv("VPBS_LEVEL_PIECES") ? std::max(1, std::atoi(std::getenv("VPBS_LEVEL_PIECES"))) : 3;
            const u64 piece = std::max<u64>(8 * 140, total / (pieces_per_thread * threads) + 1);
            std::atomic<u64> cursor{0};
            pool->share([&](unsigned t) {
#if defined(VPBS_HAVE_POSEIDON_X8)
                // the PoseidonGate rows of a piece eight at a time (batches of three or more; the rest one by one)
                u32 pending[8];
                unsigned np = 0;
                auto flush = [&] {
                    if (np >= 3) {
                        if (!poseidon_rows_x8(p, s, pending, np, errs[t])) failed.store(true);
                    } else {
                        for (unsigned q = 0; q < np; ++q)
                            if (!run_one(p, s, pending[q], mc, errs[t])) failed.store(true);
                    }
                    np = 0;
                };
#endif
                for (;;) {
                    const u64 at = cursor.fetch_add(piece, std::memory_order_relaxed);
                    if (at >= total || failed.load(std::memory_order_relaxed)) break;
                    const u32 k0 = (u32)(std::lower_bound(c0, c1, (u32)(*c0 + at)) - P.cost.data());
                    const u32 k1 = at + piece >= total ? hi : (u32)(std::lower_bound(c0, c1, (u32)(*c0 + at + piece)) - P.cost.data());
                    for (u32 k = k0; k < k1 && !failed.load(std::memory_order_relaxed); ++k) {
#if defined(VPBS_HAVE_POSEIDON_X8)
                        const auto& st = p.schedule[P.order[k]];
                        if (x8 && st.row != NO_ROW && p.gates[p.row_gate[st.row]].kind == VPBS_GATE_POSEIDON) {
                            pending[np++] = P.order[k];
                            if (np == 8) flush();
                            continue;
                        }
#endif
                        if (!run_one(p, s, P.order[k], mc, errs[t])) failed.store(true);
                    }
#if defined(VPBS_HAVE_POSEIDON_X8)
                    flush();
#endif
                }
            });
            if (failed.load()) rc = VPBS_ERR_INVALID;
            if (trace) t_wide += clock() - t_level, ++n_wide;
        }
        if (rc == VPBS_OK && s.failed()) rc = VPBS_ERR_INVALID;   // the text is read further down, when the lanes have stopped too
    }
    if (rc != VPBS_OK) failed.store(true);   // lets waiting lanes go
    const double t_lanes = trace ? clock() : 0;
    if (lane_pool) {
        lane_pool->finish();
        lane_pool->end();
    } else if (rc == VPBS_OK) {   // no threads for the lanes (another run has them): their steps here, in schedule order
        for (u32 i : p.lane_steps_sorted[ph])
            if (!run_one(p, s, i, mc, errs[0])) {
                rc = VPBS_ERR_INVALID;
                break;
            }
    }
    if (failed.load() || s.failed()) rc = VPBS_ERR_INVALID;
    if (trace && !lanes.empty()) std::fprintf(stderr, "[witness %s] %zu chain lanes (%zu PoseidonGate rows): %.2f ms after the levels\n", name, lanes.size(),
                                              p.lane_steps_sorted[ph].size(), clock() - t_lanes);
    const double t_after = trace ? clock() : 0;
    if (rc == VPBS_OK && after) {
        if (!pool_awake) pool->begin();
        pool_awake = true;
        pool->share([&](unsigned t) { after(t, threads); });
    }
    if (keep_awake && rc == VPBS_OK && awake_for) {
        if (!pool_awake) pool->begin();   // spinning by the time the next stage arrives
        promise_awake();
    } else if (pool_awake) {
        pool->end();
    }
    if (trace)
        std::fprintf(stderr, "[witness %s] %u threads: %u narrow levels %.2f ms, %u wide levels %.2f ms, wires %.2f ms\n", name, threads,
                     n_levels - n_wide, t_narrow, n_wide, t_wide, clock() - t_after);
    if (rc != VPBS_OK) {
        for (const auto& e : errs)
            if (err.empty() && !e.empty()) err = e;
        for (const auto& e : lane_errs)
            if (err.empty() && !e.empty()) err = e;
    }
    if (rc != VPBS_OK && err.empty()) {   // every thread of the run has finished: the state's message can be read
        const std::string set_twice = s.error_text();
        err = set_twice.empty() ? "a generator failed" : set_twice;
    }
    return rc;
}

// vpbs_host_set_cpu_budget: what the host says this PROCESS may use (0 = not set).  Several prover processes share one machine when every
// GPU of a node has its own rank: each of them sees the whole cgroup quota, so the launcher divides it.
std::atomic<unsigned> g_cpu_budget{0};

// CPUs this process may really use: the hardware threads, capped by the scheduler affinity, by the cgroup's CPU quota (a container on a
// 256-thread host is often given 16) and by the budget the host set
unsigned usable_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (const unsigned b = g_cpu_budget.load()) n = std::min(n, b);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
        char q[64] = {0};
        unsigned long period = 0;
        if (std::fscanf(f, "%63s %lu", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0)
            n = std::min(n, (unsigned)std::max(1ul, std::strtoul(q, nullptr, 10) / period));
        std::fclose(f);
    }
    return n;
}

std::atomic<unsigned> g_late_threads{0};   // vpbs_host_set_late_threads (0: default)
std::atomic<unsigned> g_early_threads{0};  // vpbs_host_set_early_threads (0: default)
unsigned default_phase_threads();
// The late phase's pool: VPBS_LATE_THREADS, else vpbs_host_set_late_threads, else the default of every phase.  The last late stage of an
// in-circuit verifier is 28 independent queries (strands): a host that runs ONE chain and has the CPUs asks for 14 (two queries per thread;
// measured on the 16-CPU GPU box: 8 threads 12.1 ms per chained step, 14 threads 11.5); with six chains per GPU 14 threads per chain
// cost throughput (8.64 against 8.27 ms per chained proof with 8), which is why 14 is not the default.
unsigned late_phase_threads() {
    static const char* const e_late = std::getenv("VPBS_LATE_THREADS");
    if (e_late) return (unsigned)std::max(1, atoi(e_late));
    if (const unsigned t = g_late_threads.load()) return t;
    return default_phase_threads();
}

unsigned default_phase_threads() {
    // 8 on the GPU box (256 hardware threads, 16 by its cgroup), 4 of 8.  With the PoseidonGate rows of a level generated eight per AVX-512
    // register, 6, 8 and 12 threads measure the same for one chain (the 112 rows of a query level are two batches per thread either way);
    // fewer threads leave more of a CPU quota to the other phases and chains.
    return std::max(1u, std::min(8u, usable_cpus() / 2));
}
}  // namespace
}  // namespace vpbs

extern "C" {

int vpbs_host_set_cpu_budget(unsigned cpus) {
    vpbs::g_cpu_budget.store(cpus);
    vpbs::blocking_sync_budget_changed();
    return VPBS_OK;
}
unsigned vpbs_host_cpu_budget(void) { return vpbs::usable_cpus(); }
int vpbs_host_set_late_threads(unsigned threads) {
    vpbs::g_late_threads.store(threads);
    return VPBS_OK;
}
int vpbs_host_set_early_threads(unsigned threads) {
    vpbs::g_early_threads.store(threads);
    return VPBS_OK;
}

int vpbs_witness_plan_split(vpbs_witness_plan* plan, const uint8_t* late, char* err, size_t err_len) {
    std::string msg;
    const int rc = vpbs::plan_split(plan, late, msg);
    vpbs::report(err, err_len, msg);
    return rc;
}

static int run_early_impl(const vpbs_witness_plan* plan, const uint64_t* preset_val, unsigned threads, uint64_t* wires_out, bool recycled,
                          vpbs_witness_state** state_out, char* err, size_t err_len) {
    using namespace vpbs;
    if (!plan || !plan->is_split || !preset_val || !wires_out || !state_out) {
        report(err, err_len, "malformed arguments (is the plan split?)");
        return VPBS_ERR_INVALID;
    }
    const vpbs_witness_plan& p = *plan;
    const bool trace = std::getenv("VPBS_TRACE_WITNESS") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto t1 = std::chrono::steady_clock::now();
        if (trace) std::fprintf(stderr, "[witness early] %-12s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    // values stay unwritten until a generator sets them (the pages are then first touched by the pool's threads, not zeroed here by one)
    auto* st = new vpbs_witness_state{SlotState{Buf<u64>(p.n_slots, false), Buf<uint8_t>(p.n_slots, true), p.n, {}}};
    SlotState& s = st->s;
    lap("state");
    for (size_t i = 0; i < p.preset_slot.size(); ++i)
        if (!p.preset_late[i]) s.set(p.preset_slot[i], preset_val[i], p.preset_pos[i]);
    lap("presets");
    std::string msg = s.error;
    if (threads == 0) {   // the environment variable overrides the default; the pool keeps the size of the phase's first run either way
        static const char* const e = std::getenv("VPBS_EARLY_THREADS");
        const unsigned set = g_early_threads.load();
        threads = e ? (unsigned)std::max(1, atoi(e)) : (set ? set : default_phase_threads());
    }
    // the generators, then every wire takes its class's value: the late classes are still zero and are overwritten by run_late
    auto fill = [&](unsigned t, unsigned of) {
        const size_t lo = p.total * t / of, hi = p.total * (t + 1) / of;
        if (!recycled) std::memset(wires_out + lo, 0, (hi - lo) * sizeof(u64));
        const size_t a = std::lower_bound(p.out_pos.begin(), p.out_pos.end(), (u32)lo) - p.out_pos.begin();
        for (size_t i = a; i < p.out_pos.size() && p.out_pos[i] < hi; ++i) wires_out[p.out_pos[i]] = s.is_set[p.out_slot[i]] ? s.val[p.out_slot[i]] : 0;
    };
    const int rc = msg.empty() ? run_levels(p, 0, 0, s, threads, fill, msg) : VPBS_ERR_INVALID;
    if (rc != VPBS_OK) {
        report(err, err_len, msg);
        delete st;
        return rc;
    }
    lap("generators + wires");
    prefault_late_slots(p, s);
    lap("late pages");
    *state_out = st;
    return VPBS_OK;
}

int vpbs_witness_plan_run_early(const vpbs_witness_plan* plan, const uint64_t* preset_val, unsigned threads, uint64_t* wires_out,
                                vpbs_witness_state** state_out, char* err, size_t err_len) {
    return run_early_impl(plan, preset_val, threads, wires_out, false, state_out, err, err_len);
}
int vpbs_witness_plan_run_early_recycled(const vpbs_witness_plan* plan, const uint64_t* preset_val, unsigned threads, uint64_t* wires_out,
                                         vpbs_witness_state** state_out, char* err, size_t err_len) {
    return run_early_impl(plan, preset_val, threads, wires_out, true, state_out, err, err_len);
}

static int run_late_impl(const vpbs_witness_plan* plan, vpbs_witness_state* state, const uint64_t* preset_val, uint64_t* wires_out,
                         bool packed, char* err, size_t err_len) {
    using namespace vpbs;
    if (!plan || !plan->is_split || !state || !preset_val || !wires_out) {
        report(err, err_len, "malformed arguments");
        delete state;   // consumed whatever happens (header: only a state that never reached run_late is freed by the caller)
        return VPBS_ERR_INVALID;
    }
    if (state->failed) {   // a stage that ran ahead failed: its generators did not all run, nothing later may be built on it
        const int rc_ahead = state->failed;
        report(err, err_len, "a late stage run ahead had failed: " + state->failure);
        delete state;
        return rc_ahead;
    }
    const vpbs_witness_plan& p = *plan;
    SlotState& s = state->s;
    const bool trace = std::getenv("VPBS_TRACE_WITNESS") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto t1 = std::chrono::steady_clock::now();
        if (trace) std::fprintf(stderr, "[witness late] %-12s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    const unsigned late_threads = late_phase_threads();
    // the wires of late_out[lo, hi) (a stage's share is one such range), thread t of `of`
    auto emit = [&](size_t lo, size_t hi, unsigned t, unsigned of) {
        const size_t cnt = hi - lo;
        if (packed) return pack_late_range(p, s, wires_out, lo + cnt * t / of, lo + cnt * (t + 1) / of);
        for (size_t k = lo + cnt * t / of, end = lo + cnt * (t + 1) / of; k < end; ++k) {
            const u32 slot = p.out_slot[p.late_out[k]];
            wires_out[packed ? k : p.out_pos[p.late_out[k]]] = s.is_set[slot] ? s.val[slot] : 0;
        }
    };
    // stages that ran ahead without leaving their wires in the packed buffer (a host that wants them there passes it to run_late_stage)
    const unsigned have = packed ? state->stages_packed : 0;
    if (have < state->stages_done) {
        emit(p.late_out_stage_off[have], p.late_out_stage_off[state->stages_done], 0, 1);
        lap("wires of earlier stages");
    }
    // the stages that have not run yet (all of them unless vpbs_witness_plan_run_late_stage ran some ahead), in order, each followed by its wires
    std::string msg;
    int rc = VPBS_OK;
    for (unsigned stage = state->stages_done + 1; stage <= p.n_stages && rc == VPBS_OK; ++stage) {
        const std::vector<u32>& pre = p.stage_presets[stage - 1];
        auto presets = [&](unsigned t, unsigned of) {
            for (size_t k = pre.size() * t / of, end = pre.size() * (t + 1) / of; k < end; ++k)
                s.set(p.preset_slot[pre[k]], preset_val[pre[k]], p.preset_pos[pre[k]]);
        };
        const size_t lo = p.late_out_stage_off[stage - 1], hi = p.late_out_stage_off[stage];
        rc = run_levels(p, 1, p.n_stages > 1 ? stage : 0, s, late_threads, [&](unsigned t, unsigned of) { emit(lo, hi, t, of); }, msg, presets);
        state->release_awake();   // the pool had been left spinning for this stage: asleep again now (unless another run took it meanwhile)
        lap("presets + generators + late wires");
    }
    if (trace) {
        for (int ph = 0; ph < 2; ++ph) {
            const vpbs_witness_plan::Phase& P = p.phase[ph];
            size_t n_pos = 0, wide = 0, wide_steps = 0, wide_pos = 0, narrow_pos = 0;
            for (size_t l = 0; l + 1 < P.level_off.size(); ++l) {
                size_t pos_rows = 0;
                for (u32 k = P.level_off[l]; k < P.level_off[l + 1]; ++k) {
                    const auto& st = p.schedule[P.order[k]];
                    if (st.row != NO_ROW && p.gates[p.row_gate[st.row]].kind == VPBS_GATE_POSEIDON) ++pos_rows;
                }
                n_pos += pos_rows;
                if (P.cost[P.level_off[l + 1]] - P.cost[P.level_off[l]] >= 600) {
                    ++wide;
                    wide_steps += P.level_off[l + 1] - P.level_off[l];
                    wide_pos += pos_rows;
                } else {
                    narrow_pos += pos_rows;
                }
            }
            std::fprintf(stderr, "[witness %s] %zu generators (%zu PoseidonGate rows) of %zu; %zu levels, %zu of them wide with %zu generators "
                         "(%zu PoseidonGate rows; %zu PoseidonGate rows sit in narrow levels)\n", ph ? "late" : "early", P.order.size(), n_pos,
                         p.schedule.size(), P.level_off.size() - 1, wide, wide_steps, wide_pos, narrow_pos);
        }
        std::fprintf(stderr, "[witness late] %zu late wire positions\n", p.late_out.size());
    }
    report(err, err_len, msg);
    delete state;
    return rc;
}

int vpbs_witness_plan_run_late(const vpbs_witness_plan* plan, vpbs_witness_state* state, const uint64_t* preset_val, uint64_t* wires_out,
                               char* err, size_t err_len) {
    return run_late_impl(plan, state, preset_val, wires_out, false, err, err_len);
}
int vpbs_witness_plan_run_late_packed(const vpbs_witness_plan* plan, vpbs_witness_state* state, const uint64_t* preset_val,
                                      uint64_t* values_out, char* err, size_t err_len) {
    return run_late_impl(plan, state, preset_val, values_out, true, err, err_len);
}
unsigned vpbs_witness_plan_late_stages(const vpbs_witness_plan* plan) { return plan && plan->is_split ? plan->n_stages : 0; }

int vpbs_witness_plan_run_late_stage(const vpbs_witness_plan* plan, vpbs_witness_state* state, unsigned stage, const uint64_t* preset_val,
                                     uint64_t* packed_out, char* err, size_t err_len) {
    using namespace vpbs;
    if (!plan || !plan->is_split || !state || !preset_val || stage == 0 || stage > plan->n_stages || stage != state->stages_done + 1) {
        report(err, err_len, "malformed arguments (stages run once each, in ascending order)");
        return VPBS_ERR_INVALID;
    }
    if (state->failed) {
        report(err, err_len, "an earlier late stage had failed: " + state->failure);
        return state->failed;
    }
    const vpbs_witness_plan& p = *plan;
    SlotState& s = state->s;
    const std::vector<u32>& pre = p.stage_presets[stage - 1];
    auto presets = [&](unsigned t, unsigned of) {
        for (size_t k = pre.size() * t / of, end = pre.size() * (t + 1) / of; k < end; ++k)
            s.set(p.preset_slot[pre[k]], preset_val[pre[k]], p.preset_pos[pre[k]]);
    };
    std::string msg;
    const unsigned late_threads = late_phase_threads();
    // with a packed buffer the stage leaves its share of the late wires there at once (the order of vpbs_witness_plan_late_positions is by stage)
    std::function<void(unsigned, unsigned)> after;
    const bool pack = packed_out && state->stages_packed + 1 == stage;
    if (pack) {
        const size_t lo = p.late_out_stage_off[stage - 1], hi = p.late_out_stage_off[stage];
        after = [&p, &s, packed_out, lo, hi](unsigned t, unsigned of) {
            const size_t cnt = hi - lo;
            pack_late_range(p, s, packed_out, lo + cnt * t / of, lo + cnt * (t + 1) / of);
        };
    }
    // the stage before the last leaves the pool spinning: the last one starts when the proof returns, a fraction of a millisecond later, and
    // waking sleeping workers would cost it more than that
    state->release_awake();
    const bool keep = stage + 1 == p.n_stages;
    const int rc = run_levels(p, 1, p.n_stages > 1 ? stage : 0, s, late_threads, after, msg, presets, keep, keep ? state : nullptr);
    if (pack && rc == VPBS_OK) state->stages_packed = stage;
    state->stages_done = stage;
    if (rc != VPBS_OK) {          // the state is only good for vpbs_witness_state_free / run_late now, and both know: the failure is the state's
        state->failed = rc;
        state->failure = msg.empty() ? "stage " + std::to_string(stage) + " failed" : msg;
    }
    report(err, err_len, msg);
    return rc;
}

size_t vpbs_witness_plan_late_count(const vpbs_witness_plan* plan) { return plan && plan->is_split ? plan->late_out.size() : 0; }
int vpbs_witness_plan_late_positions(const vpbs_witness_plan* plan, uint32_t* out) {
    if (!plan || !plan->is_split || !out) return VPBS_ERR_INVALID;
    for (size_t k = 0; k < plan->late_out.size(); ++k) out[k] = (uint32_t)plan->out_pos[plan->late_out[k]];
    return VPBS_OK;
}

size_t vpbs_witness_plan_late_input_count(const vpbs_witness_plan* plan) { return plan && plan->is_split ? plan->late_in_slots.size() : 0; }
int vpbs_witness_plan_late_input_positions(const vpbs_witness_plan* plan, uint32_t* out) {
    if (!plan || !plan->is_split || !out) return VPBS_ERR_INVALID;
    // one wire position per slot: the first of its copy class (out_pos is ascending)
    std::vector<vpbs::u32> first(plan->n_slots, 0xFFFFFFFFu);
    for (size_t i = plan->out_slot.size(); i-- > 0;) first[plan->out_slot[i]] = plan->out_pos[i];
    for (size_t k = 0; k < plan->late_in_slots.size(); ++k) out[k] = first[plan->late_in_slots[k]];
    return VPBS_OK;
}
int vpbs_witness_state_from_late_inputs(const vpbs_witness_plan* plan, const uint64_t* values, vpbs_witness_state** state_out) {
    using namespace vpbs;
    if (!plan || !plan->is_split || !state_out || (!values && !plan->late_in_slots.empty())) return VPBS_ERR_INVALID;
    const vpbs_witness_plan& p = *plan;
    auto* st = new vpbs_witness_state{SlotState{Buf<u64>(p.n_slots, false), Buf<uint8_t>(p.n_slots, true), p.n, {}}};
    for (size_t k = 0; k < p.late_in_slots.size(); ++k) {
        if (values[k] >= gl::P) {
            delete st;
            return VPBS_ERR_INVALID;
        }
        st->s.val[p.late_in_slots[k]] = values[k];
        st->s.is_set[p.late_in_slots[k]] = 1;
    }
    prefault_late_slots(p, st->s);
    *state_out = st;
    return VPBS_OK;
}

void vpbs_witness_state_free(vpbs_witness_state* state) { delete state; }

int vpbs_witness_plan_late_rows(const vpbs_witness_plan* plan, size_t out[2]) {
    if (!plan || !plan->is_split || !out) return VPBS_ERR_INVALID;
    size_t lo = plan->n, hi = 0;
    for (vpbs::u32 i : plan->late_out) {
        const size_t r = plan->out_pos[i] % plan->n;
        lo = std::min(lo, r);
        hi = std::max(hi, r + 1);
    }
    out[0] = hi ? lo : 0;
    out[1] = hi;
    return VPBS_OK;
}

int vpbs_gate_fill_row(const vpbs_gate* gp, const uint64_t* constants, uint64_t* row) {
    if (!gp || !row) return VPBS_ERR_INVALID;
    vpbs_gate g = *gp;
    if (vpbs_gate_default_params(&g) != VPBS_OK) return VPBS_ERR_INVALID;
    if (g.num_constants && !constants) return VPBS_ERR_INVALID;
    vpbs::PlainRow r{row};
    try {
        for (unsigned sub = 0; sub < vpbs::gen_count(g); ++sub) vpbs::gen_run(g, sub, constants, r, vpbs::tables_of(g));
    } catch (const vpbs::GenError&) {
        return VPBS_ERR_INVALID;
    }
    return VPBS_OK;
}

int vpbs_selector_columns(const vpbs_circuit* c, uint64_t* out) {
    if (!vpbs::check_circuit(c) || !out) return VPBS_ERR_INVALID;
    const size_t n = (size_t)1 << c->log_n;
    for (unsigned s = 0; s < c->num_selectors; ++s)
        for (size_t r = 0; r < n; ++r) {
            const vpbs_gate& g = c->gates[c->row_gate[r]];
            out[s * n + r] = g.selector_index == s ? g.index : VPBS_UNUSED_SELECTOR;
        }
    return VPBS_OK;
}

int vpbs_sigma_values(const vpbs_circuit* c, uint64_t* out) {
    if (!vpbs::check_circuit(c) || !out) return VPBS_ERR_INVALID;
    using namespace vpbs;
    const size_t n = (size_t)1 << c->log_n, total = (size_t)c->n_routed * n;
    Partition part(total);
    for (size_t i = 0; i < c->n_copies; ++i) part.merge(c->copies[2 * i], c->copies[2 * i + 1]);
    // subsets in WirePartition order: members sorted by (row, column); sigma maps each member to the next, cyclically
    std::vector<u32> order(total);
    std::iota(order.begin(), order.end(), 0u);
    auto key = [&](u32 p) { return ((u64)(p % n) << 32) | (u64)(p / n); };
    std::vector<std::pair<u64, u32>> tagged;  // (representative, position) for positions in non-trivial subsets
    std::vector<u32> size(total, 0);
    for (u32 p = 0; p < total; ++p) ++size[part.find(p)];
    for (u32 p = 0; p < total; ++p)
        if (size[part.find(p)] > 1) tagged.push_back({part.find(p), p});
    std::sort(tagged.begin(), tagged.end(), [&](const auto& a, const auto& b) {
        return a.first != b.first ? a.first < b.first : key(a.second) < key(b.second);
    });
    std::vector<u32> sigma(total);
    std::iota(sigma.begin(), sigma.end(), 0u);
    for (size_t i = 0; i < tagged.size();) {
        size_t j = i;
        while (j < tagged.size() && tagged[j].first == tagged[i].first) ++j;
        for (size_t k = i; k < j; ++k) sigma[tagged[k].second] = tagged[k + 1 < j ? k + 1 : i].second;
        i = j;
    }
    std::vector<u64> subgroup(n), k_is(c->n_routed);
    const u64 w = gl::root_of_unity(c->log_n);
    u64 x = 1;
    for (size_t i = 0; i < n; ++i, x = gl::mul(x, w)) subgroup[i] = x;
    x = 1;
    for (unsigned j = 0; j < c->n_routed; ++j, x = gl::mul(x, gl::GENERATOR)) k_is[j] = x;
    for (size_t p = 0; p < total; ++p) out[p] = gl::mul(k_is[sigma[p] / n], subgroup[sigma[p] % n]);
    return VPBS_OK;
}

int vpbs_witness_plan_create(const vpbs_circuit* c, const uint32_t* preset_pos, size_t n_preset, vpbs_witness_plan** out, char* err,
                             size_t err_len) {
    std::string msg;
    const int rc = vpbs::plan_create(c, preset_pos, n_preset, out, msg);
    vpbs::report(err, err_len, msg);
    return rc;
}

int vpbs_witness_plan_run(const vpbs_witness_plan* plan, const uint64_t* preset_val, unsigned threads, uint64_t* wires_out, char* err,
                          size_t err_len) {
    std::string msg;
    const int rc = vpbs::plan_run(plan, preset_val, threads, wires_out, msg);
    vpbs::report(err, err_len, msg);
    return rc;
}

void vpbs_witness_plan_free(vpbs_witness_plan* plan) { delete plan; }

int vpbs_witness_plan_stats(const vpbs_witness_plan* plan, uint64_t out[4]) {
    if (!plan || !out) return VPBS_ERR_INVALID;
    out[0] = plan->n_slots;
    out[1] = plan->schedule.size();
    out[2] = plan->dev.supported ? plan->dev.n_levels : 0;
    out[3] = plan->out_pos.size();
    return VPBS_OK;
}

int vpbs_generate_witness(const vpbs_circuit* c, const uint32_t* preset_pos, const uint64_t* preset_val, size_t n_preset,
                          uint64_t* wires_out, char* err, size_t err_len) {
    std::string msg;
    vpbs_witness_plan* plan = nullptr;
    int rc = wires_out ? vpbs::plan_create(c, preset_pos, n_preset, &plan, msg) : VPBS_ERR_INVALID;
    if (rc == VPBS_OK) rc = vpbs::plan_run(plan, preset_val, 0, wires_out, msg);
    delete plan;
    if (rc != VPBS_OK && msg.empty()) msg = "malformed circuit description";
    vpbs::report(err, err_len, msg);
    return rc;
}

int vpbs_check_witness(const vpbs_circuit* c, const uint64_t* wires, const uint64_t pi_hash[4], char* err, size_t err_len) {
    auto report = [&](const std::string& m) {
        if (err && err_len) {
            std::strncpy(err, m.c_str(), err_len - 1);
            err[err_len - 1] = 0;
        }
    };
    if (!vpbs::check_circuit(c) || !wires || !pi_hash) {
        report("malformed circuit description");
        return VPBS_ERR_INVALID;
    }
    using namespace vpbs;
    const size_t n = (size_t)1 << c->log_n;
    unsigned max_consts = 0;
    for (unsigned i = 0; i < c->n_gates; ++i) max_consts = std::max(max_consts, c->gates[i].num_constants);
    if (c->num_selectors + max_consts > c->n_constants_cols || (max_consts && !c->constants)) {
        report("constants columns missing");
        return VPBS_ERR_INVALID;
    }
    struct RowVars {
        const u64* wires;
        const u64* constants;
        size_t n, row;
        unsigned n_wires, first_const, n_const_cols;
        const u64* pih;
        u64 wire(unsigned i) const { return i < n_wires ? wires[(size_t)i * n + row] : 0; }
        u64 constant(unsigned i) const { return first_const + i < n_const_cols ? constants[(size_t)(first_const + i) * n + row] : 0; }
        u64 pi_hash(unsigned i) const { return pih[i]; }
    };
    struct Sink {
        std::vector<u64> c;
        void push(u64 x) { c.push_back(x); }
    };
    for (size_t r = 0; r < n; ++r) {
        const vpbs_gate& g = c->gates[c->row_gate[r]];
        if (g.num_constraints == 0) continue;
        RowVars v{wires, c->constants, n, r, c->n_wires, c->num_selectors, c->n_constants_cols, pi_hash};
        Sink s;
        const gates::CosetTables* t = g.kind == VPBS_GATE_COSET_INTERPOLATION ? &gates::coset_tables(g.p0) : nullptr;
        gates::eval_gate<u64>(g, t, v, s);
        for (size_t k = 0; k < s.c.size(); ++k)
            if (s.c[k] != 0) {
                char id[256] = "gate";
                (void)vpbs_gate_id(&g, id, sizeof id);
                id[60] = 0;
                report("row " + std::to_string(r) + ": constraint " + std::to_string(k) + " of " + id + " is not satisfied");
                return 0;
            }
    }
    for (size_t i = 0; i < c->n_copies; ++i) {
        const uint32_t a = c->copies[2 * i], b = c->copies[2 * i + 1];
        if (wires[a] != wires[b]) {
            report("copy constraint violated: (column " + std::to_string(a / n) + ", row " + std::to_string(a % n) + ") != (column " +
                   std::to_string(b / n) + ", row " + std::to_string(b % n) + ")");
            return 0;
        }
    }
    if (err && err_len) err[0] = 0;
    return 1;
}

}  // extern "C"
