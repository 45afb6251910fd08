// engine_rules.hip -- reward rules of the host engine: which rule shapes the kernels take (RuleArgs / RuleProg), and the reference's recursive
// search on the host for the others (RewardEngine.cc:105-443)
#include "engine_impl.h"

namespace magent_amd {

// A rule shape the GPU kernels do not take (several iterated symbols, 'all' / fixed-index symbols, in_a_line, receivers in
// the subject's and the object's group at once ...) is not refused: ALL rules of such a game are evaluated on the host by
// the reference's recursive search (Env::eval_rules_host) -- all of them, because the float adds of different rules on
// one agent have to keep their order.
namespace { struct RuleGoesToHost { char why[256]; }; }
[[noreturn]] static void to_host(const char *fmt, ...) {
    RuleGoesToHost e;
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(e.why, sizeof(e.why), fmt, ap);
    va_end(ap);
    throw e;
}

// translate the rule shapes the kernels take into kernel arguments; the other shapes send the game's rules to the host
void Env::compile_rules() {
    rule_args.clear();
    rule_progs.clear();
    rules_on_host = false;
    try {
        compile_rules_gpu();
    } catch (const RuleGoesToHost &e) {
        rules_on_host = true;
        rule_args.clear();
        rule_progs.clear();
        if (std::getenv("MAGENT_VERBOSE")) std::fprintf(stderr, "magent-amd: reward rules evaluated on the host (%s)\n", e.why);
    }
    if (rules_on_host) plan_host_rules();
}

void Env::compile_rules_gpu() {
    if ((int)rules.size() > CTR_TRIGGER_END - CTR_TRIGGER) to_host("too many reward rules");
    for (size_t k = 0; k < rules.size(); k++) {
        const HostRule &r = rules[k];
        if (r.on < 0 || r.on >= (int)nodes.size()) fatal("reward rule %zu refers to an undefined event", k);
        const HostNode &on = nodes[r.on];
        auto binary = [&](const HostNode &n) { return (n.op == OP_ATTACK || n.op == OP_KILL || n.op == OP_COLLIDE) && n.raw.size() == 2; };
        auto any_sym = [&](int no) {
            if (no < 0 || no >= (int)symbols.size()) fatal("reward rule %zu refers to an undefined agent symbol", k);
            const HostSymbol &sy = symbols[no];
            if (sy.index != -1) to_host("reward rule %zu: only 'any' agent symbols are on the GPU path", k);
            if (sy.group < 0 || sy.group >= (int)groups.size()) fatal("reward rule %zu: invalid group in agent symbol", k);
            return sy.group;
        };
        if (on.op == 0 /* and */ && on.raw.size() == 2 && on.raw[0] >= 0 && on.raw[1] >= 0 && on.raw[0] < (int)nodes.size() &&
            on.raw[1] < (int)nodes.size() && binary(nodes[on.raw[0]]) && binary(nodes[on.raw[1]])) {
            // Event(a, p, c) & Event(b, q, c): "two agents act on the same third" (builtin/config/double_attack.py:33-40)
            const HostNode *e1 = &nodes[on.raw[0]], *e2 = &nodes[on.raw[1]];
            if (e1->raw[1] != e2->raw[1] || e1->raw[0] == e2->raw[0] || e1->raw[0] == e1->raw[1] || e2->raw[0] == e2->raw[1]) {
                compile_rule_program(k);   // not "two agents on one object": a general expression, if its search iterates one symbol
                continue;
            }
            if (e2->raw[0] < e1->raw[0]) std::swap(e1, e2);   // the search binds symbols in ascending number (RewardEngine.cc:155-189)
            RuleArgs a{};
            a.prog = -1;
            a.pair = 1; a.rule_no = (int)k;
            a.ga = any_sym(e1->raw[0]); a.op = e1->op;
            a.gy = any_sym(e2->raw[0]); a.op_y = e2->op;
            a.gb = any_sym(e1->raw[1]);
            for (size_t i = 0; i < r.recv.size(); i++) {
                int *cnt; float *val;
                if (r.recv[i] == e1->raw[0]) { cnt = &a.n_subj; val = a.v_subj; }
                else if (r.recv[i] == e2->raw[0]) { cnt = &a.n_y; val = a.v_y; }
                else if (r.recv[i] == e1->raw[1]) { cnt = &a.n_obj; val = a.v_obj; }
                else to_host("reward rule %zu: a receiver must be a symbol of the event", k);
                if (*cnt == 4) to_host("too many receivers");
                val[(*cnt)++] = r.val[i];
            }
            if (a.n_obj && (a.gb == a.ga || a.gb == a.gy))
                to_host("reward rule %zu: paying the shared object inside a subject's group interleaves float adds; not on the GPU path", k);
            rule_args.push_back(a);
            continue;
        }
        if (!binary(on)) {   // a general expression: on the GPU path when its search iterates a single symbol
            compile_rule_program(k);
            continue;
        }
        const int group_a = any_sym(on.raw[0]), group_b = any_sym(on.raw[1]);
        // one symbol as subject AND object: the reference binds the object over the subject's entity (RewardEngine.cc:17-24,
        // 405-408) and then tests the TARGET against itself -- it fires for an agent whose target hit itself (bodies whose
        // in-group attack range covers their own cells): the program form knows that shape
        if (on.raw[0] == on.raw[1]) { compile_rule_program(k); continue; }
        RuleArgs a{};
        a.prog = -1;
        a.ga = group_a; a.gb = group_b; a.op = on.op; a.rule_no = (int)k;
        for (size_t i = 0; i < r.recv.size(); i++) {
            if (r.recv[i] == on.raw[0]) { if (a.n_subj == 4) to_host("too many receivers"); a.v_subj[a.n_subj++] = r.val[i]; }
            else if (r.recv[i] == on.raw[1]) { if (a.n_obj == 4) to_host("too many receivers"); a.v_obj[a.n_obj++] = r.val[i]; }
            else to_host("reward rule %zu: a receiver must be the subject or the object of the event", k);
        }
        if (a.n_subj && a.n_obj && a.ga == a.gb)
            to_host("reward rule %zu: subject and object receivers in the same group interleave float adds; not on the GPU path", k);
        rule_args.push_back(a);
    }
}

// A rule whose event is a general expression (and / or / not over attack, kill, collide, die, at, in).  The reference
// plans its search per rule (GridWorld::init_reward_description, RewardEngine.cc:105-214): the symbols of the expression
// in ascending number; a symbol that is the subject of a binary event brings that event's object along ("inferred":
// bound to the subject's op_obj instead of being iterated).  The GPU path takes the rules whose plan iterates ONE
// symbol -- every other symbol is its inferred object -- and evaluates the expression per agent (k_rule_prog).
void Env::compile_rule_program(size_t k) {
    const HostRule &r = rules[k];
    struct Info { std::vector<int> related; std::vector<std::pair<int, int>> infer; };
    std::function<Info(int)> collect = [&](int no) -> Info {
        if (no < 0 || no >= (int)nodes.size()) fatal("reward rule %zu refers to an undefined event", k);
        const HostNode &n = nodes[no];
        Info I;
        auto add_sym = [&](int s2) { if (std::find(I.related.begin(), I.related.end(), s2) == I.related.end()) I.related.push_back(s2); };
        auto add_inf = [&](std::pair<int, int> p) { for (auto &q : I.infer) if (q.first == p.first) return; I.infer.push_back(p); };
        if (n.op == 0 || n.op == 1 || n.op == 2) {
            const size_t kids = n.op == 2 ? 1 : 2;
            if (n.raw.size() < kids) fatal("reward rule %zu: malformed event node", k);
            for (size_t c = 0; c < kids; c++) {
                Info C = collect(n.raw[c]);
                for (int s2 : C.related) add_sym(s2);
                for (auto &p : C.infer) add_inf(p);
            }
        } else if (n.op == OP_KILL || n.op == OP_COLLIDE || n.op == OP_ATTACK) {
            add_sym(n.raw[0]); add_sym(n.raw[1]); add_inf({n.raw[0], n.raw[1]});
        } else if (n.op == 4 || n.op == 5 || n.op == 8) {   // at, in, die
            add_sym(n.raw[0]);
        } else to_host("reward rule %zu: event predicate %d (in_a_line / align) is not on the GPU path", k, n.op);
        std::sort(I.related.begin(), I.related.end());
        std::sort(I.infer.begin(), I.infer.end());
        return I;
    };
    const Info I = collect(r.on);
    std::vector<int> iterated, inferred, added;
    auto has = [&](int s2) { return std::find(added.begin(), added.end(), s2) != added.end(); };
    for (int s2 : I.related) {
        if (has(s2)) continue;
        for (auto &p : I.infer) if (p.first == s2) { iterated.push_back(s2); inferred.push_back(p.second); added.push_back(s2); added.push_back(p.second); break; }
    }
    for (int s2 : I.related) if (!has(s2)) { iterated.push_back(s2); inferred.push_back(-1); }
    if (iterated.size() != 1)
        to_host("reward rule %zu: its search iterates %zu agent symbols; the GPU path takes rules that iterate one symbol "
              "(plus Event(a, p, c) & Event(b, q, c))", k, iterated.size());
    // (sy == sx: an event whose subject is its own object.  The search iterates the symbol and then re-binds it to the
    // iterated agent's op_obj: every leaf and every receiver then means that target -- slot 1)
    const int sx = iterated[0], sy = inferred[0];
    const bool self = sy == sx;
    auto group_of = [&](int no) {
        if (no < 0 || no >= (int)symbols.size()) to_host("reward rule %zu refers to an undefined agent symbol", k);
        if (symbols[no].index != -1) to_host("reward rule %zu: only 'any' agent symbols are on the GPU path", k);
        if (symbols[no].group < 0 || symbols[no].group >= (int)groups.size()) fatal("reward rule %zu: invalid group in agent symbol", k);
        return symbols[no].group;
    };
    RuleProg P{};
    P.ga = group_of(sx); P.has_obj = sy >= 0; P.gb = sy >= 0 ? group_of(sy) : 0; P.rule_no = (int)k;
    auto slot = [&](int no) { if (no == sy && (self || no != sx)) return 1; if (no == sx) return 0; fatal("reward rule %zu: internal: unplanned symbol", k); return 0; };
    std::function<void(int)> emit = [&](int no) {
        const HostNode &n = nodes[no];
        if (n.op == 0 || n.op == 1) { emit(n.raw[0]); emit(n.raw[1]); }
        else if (n.op == 2) emit(n.raw[0]);
        if (P.n == 24) to_host("reward rule %zu: expression too long", k);
        P.op[P.n] = n.op;
        if (n.op == OP_KILL || n.op == OP_COLLIDE || n.op == OP_ATTACK) { P.a[P.n][0] = slot(n.raw[0]); P.a[P.n][1] = slot(n.raw[1]); }
        else if (n.op == 4 || n.op == 5 || n.op == 8) {
            P.a[P.n][0] = slot(n.raw[0]);
            const size_t want = n.op == 4 ? 3 : n.op == 5 ? 5 : 1;
            if (n.raw.size() < want) fatal("reward rule %zu: malformed event node", k);
            for (size_t q = 1; q < want; q++) P.a[P.n][q] = n.raw[q];
        }
        P.n++;
    };
    emit(r.on);
    RuleArgs a{};
    a.prog = (int)rule_progs.size(); a.rule_no = (int)k; a.ga = P.ga; a.gb = P.gb;
    for (size_t i = 0; i < r.recv.size(); i++) {
        if (r.recv[i] == sx && !self) { if (P.n_subj == 4) to_host("too many receivers"); P.v_subj[P.n_subj++] = r.val[i]; }
        else if (r.recv[i] == sy && sy >= 0) { if (P.n_obj == 4) to_host("too many receivers"); P.v_obj[P.n_obj++] = r.val[i]; }
        else to_host("reward rule %zu: a receiver must be a symbol of the event", k);
    }
    if (P.n_subj && P.n_obj && P.ga == P.gb)
        to_host("reward rule %zu: subject and object receivers in the same group interleave float adds; not on the GPU path", k);
    a.n_obj = P.n_obj;
    for (int q = 0; q < P.n_obj; q++) a.v_obj[q] = P.v_obj[q];
    rule_progs.push_back(P);
    rule_args.push_back(a);
}

// ------------------------------------------------------------------------------------------------ rules on the host
// GridWorld::init_reward_description (RewardEngine.cc:105-214): per rule, the order in which the recursive search binds
// the symbols of its event expression -- ascending symbol number; a symbol that is the subject of a binary event brings
// that event's object along (bound to the subject's op_obj instead of being iterated; the first such pair per subject,
// children left to right).
void Env::plan_host_rules() {
    struct Info { std::vector<int> related; std::vector<std::pair<int, int>> infer; };
    host_plans.assign(rules.size(), HostRulePlan{});
    for (size_t k = 0; k < rules.size(); k++) {
        std::function<Info(int)> collect = [&](int no) -> Info {
            if (no < 0 || no >= (int)nodes.size()) fatal("reward rule %zu refers to an undefined event", k);
            const HostNode &n = nodes[no];
            Info I;
            auto sym = [&](int s2) {
                if (s2 < 0 || s2 >= (int)symbols.size()) fatal("reward rule %zu refers to an undefined agent symbol", k);
                if (symbols[s2].group < 0 || symbols[s2].group >= (int)groups.size()) fatal("reward rule %zu: invalid group in agent symbol", k);
                if (std::find(I.related.begin(), I.related.end(), s2) == I.related.end()) I.related.push_back(s2);
            };
            auto inf = [&](std::pair<int, int> p) { for (auto &q : I.infer) if (q.first == p.first) return; I.infer.push_back(p); };
            const size_t want = (n.op == 0 || n.op == 1) ? 2 : n.op == 4 ? 3 : n.op == 5 ? 5 : (n.op == OP_KILL || n.op == OP_COLLIDE || n.op == OP_ATTACK) ? 2 : 1;
            if (n.raw.size() < want) fatal("reward rule %zu: malformed event node", k);
            switch (n.op) {
                case 0: case 1: case 2:
                    for (size_t c = 0; c < (n.op == 2 ? 1u : 2u); c++) {
                        Info C = collect(n.raw[c]);
                        for (int s2 : C.related) sym(s2);
                        for (auto &q : C.infer) inf(q);
                    }
                    break;
                case OP_KILL: case OP_COLLIDE: case OP_ATTACK:
                    sym(n.raw[0]); sym(n.raw[1]); inf({n.raw[0], n.raw[1]});
                    if (symbols[n.raw[1]].index == -2) fatal("reward rule %zu: the object of attack / kill / collide cannot be a whole group (the reference asserts)", k);
                    break;
                case 4: case 5: case 8: sym(n.raw[0]); break;
                case 9:    // in_a_line: a statement about a whole group (the reference asserts is_all)
                    sym(n.raw[0]);
                    if (symbols[n.raw[0]].index != -2) fatal("reward rule %zu: in_a_line takes an 'all' symbol (the reference asserts)", k);
                    break;
                case 10:
                    fatal("reward rule %zu: 'align' reads two counters the reference allocates and never fills (GridWorld.cc:94-95, 955-968): "
                          "it has no defined result to reproduce", k);
                default: fatal("reward rule %zu: invalid event predicate %d", k, n.op);
            }
            std::sort(I.related.begin(), I.related.end());
            std::sort(I.infer.begin(), I.infer.end());
            return I;
        };
        const Info I = collect(rules[k].on);
        HostRulePlan &P = host_plans[k];
        std::vector<int> added;
        auto has = [&](int s2) { return std::find(added.begin(), added.end(), s2) != added.end(); };
        for (int s2 : I.related) {
            if (has(s2)) continue;
            for (auto &q : I.infer)
                if (q.first == s2) { P.order.push_back(s2); P.brings.push_back(q.second); added.push_back(s2); added.push_back(q.second); break; }
        }
        for (int s2 : I.related) if (!has(s2)) { P.order.push_back(s2); P.brings.push_back(-1); }
        for (int rc : rules[k].recv) if (rc < 0 || rc >= (int)symbols.size()) fatal("reward rule %zu: undefined receiver", k);
    }
}

// GridWorld::calc_reward with calc_rule / calc_event_node (GridWorld.cc:681-692, RewardEngine.cc:216-443) on host copies
// of what the rules read -- last_op, op_obj, positions, dead flags -- in the reference's binding order, so that every
// float add lands in the reference's order.  Rewards go back to the device; triggers stay here.
void Env::eval_rules_host() {
    const int NG = (int)groups.size();
    struct Copy { std::vector<unsigned char> last_op, dead, busy; std::vector<int> op_obj, x, y; std::vector<float> reward; bool dirty = false; };
    std::vector<Copy> C(NG);
    HIP_OK(hipStreamSynchronize(stream));
    for (int g = 0; g < NG; g++) {
        const int n = groups[g].n;
        Copy &c = C[g];
        c.last_op.resize(n); c.dead.resize(n); c.busy.assign(n, 0); c.op_obj.resize(n); c.x.resize(n); c.y.resize(n); c.reward.resize(n);
        if (!n) continue;
        const GroupDev &D = groups[g].cur;
        HIP_OK(hipMemcpy(c.last_op.data(), D.last_op, n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(c.dead.data(), D.dead, n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(c.op_obj.data(), D.op_obj, sizeof(int) * n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(c.x.data(), D.x, sizeof(int) * n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(c.y.data(), D.y, sizeof(int) * n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(c.reward.data(), D.next_reward, sizeof(float) * n, hipMemcpyDeviceToHost));
    }
    auto bound = [&](const HostSymbol &sy, size_t k) {
        if (sy.ent_g < 0 || sy.ent_g >= NG || sy.ent_i < 0 || sy.ent_i >= groups[sy.ent_g].n)
            fatal("reward rule %zu reads an agent symbol that no search has bound (the reference follows a dangling pointer here)", k);
    };
    // AgentSymbol::bind_with_check (RewardEngine.cc:14-24)
    // (Agent::index is 0 from the constructor and only clear_dead sets it, GridWorld.h:136 / GridWorld.cc:655: an agent added
    // since the last clear_dead answers 0 here, whatever its position in the group -- HostGroup::indexed)
    auto bind = [&](HostSymbol &sy, int ref) {
        if (sy.group != ref_group(ref)) return false;
        const int stored = ref_index(ref) < groups[ref_group(ref)].indexed ? ref_index(ref) : 0;
        if (sy.index != -1 && sy.index != stored) return false;
        sy.ent_g = ref_group(ref); sy.ent_i = ref_index(ref);
        return true;
    };
    host_triggers.assign(rules.size(), 0);
    for (size_t k = 0; k < rules.size(); k++) {
        const HostRule &R = rules[k];
        const HostRulePlan &P = host_plans[k];
        std::function<bool(int)> holds = [&](int no) -> bool {
            const HostNode &n = nodes[no];
            if (n.op == 0) return holds(n.raw[0]) && holds(n.raw[1]);
            if (n.op == 1) return holds(n.raw[0]) || holds(n.raw[1]);
            if (n.op == 2) return !holds(n.raw[0]);
            const HostSymbol &s0 = symbols[n.raw[0]];
            const Copy &c0 = C[s0.group];
            const int n0 = groups[s0.group].n;
            if (n.op == 9) {   // in_a_line: one column (or one row) of consecutive cells, in any order (RewardEngine.cc:263-296)
                if (n0 < 2) return true;
                const int dx = c0.x[0] - c0.x[1], dy = c0.y[0] - c0.y[1];
                if ((dx == 0) == (dy == 0)) return false;
                const std::vector<int> &fixed = dx == 0 ? c0.x : c0.y, &runs = dx == 0 ? c0.y : c0.x;
                int lo = runs[0], hi = runs[0];
                bool in_line = true;
                for (int i = 1; i < n0 && in_line; i++) { lo = std::min(lo, runs[i]); hi = std::max(hi, runs[i]); in_line = fixed[i] == fixed[0]; }
                return in_line && hi - lo + 1 == n0;
            }
            std::function<bool(int, int)> leaf;
            if (n.op == OP_KILL || n.op == OP_COLLIDE || n.op == OP_ATTACK) {
                const HostSymbol &s1 = symbols[n.raw[1]];
                bound(s1, k);
                const int obj = ref_pack(s1.ent_g, s1.ent_i);
                leaf = [&C, &n, obj](int g, int i) { return C[g].last_op[i] == n.op && C[g].op_obj[i] == obj; };
            } else if (n.op == 8) leaf = [&C](int g, int i) { return C[g].dead[i] != 0; };
            else if (n.op == 4) leaf = [&C, &n](int g, int i) { return C[g].x[i] == n.raw[1] && C[g].y[i] == n.raw[2]; };
            else leaf = [&C, &n](int g, int i) { return C[g].x[i] > n.raw[1] && C[g].x[i] < n.raw[3] && C[g].y[i] > n.raw[2] && C[g].y[i] < n.raw[4]; };
            if (s0.index == -2) {      // 'all': every agent of the group
                for (int i = 0; i < n0; i++) if (!leaf(s0.group, i)) return false;
                return true;
            }
            bound(s0, k);
            return leaf(s0.ent_g, s0.ent_i);
        };
        std::function<void(size_t)> search = [&](size_t depth) {
            if (depth == P.order.size()) {
                if (!holds(R.on)) return;
                host_triggers[k] = 1;
                for (size_t q = 0; q < R.recv.size(); q++) {
                    const HostSymbol &sy = symbols[R.recv[q]];
                    if (sy.index == -2) groups[sy.group].group_reward += R.val[q];        // Group::add_reward
                    else { bound(sy, k); C[sy.ent_g].reward[sy.ent_i] += R.val[q]; C[sy.ent_g].dirty = true; }
                }
                return;
            }
            HostSymbol &sy = symbols[P.order[depth]];
            const int brings = P.brings[depth];
            Copy &c = C[sy.group];
            const int n = groups[sy.group].n;
            if (sy.index == -1) {          // 'any': every agent of the group that no outer level of this search holds
                for (int i = 0; i < n; i++) {
                    sy.ent_g = sy.group; sy.ent_i = i;
                    if (c.busy[i]) continue;
                    c.busy[i] = 1;
                    if (brings < 0) search(depth + 1);
                    else if (c.op_obj[i] >= 0 && bind(symbols[brings], c.op_obj[i])) search(depth + 1);
                    c.busy[i] = 0;
                }
            } else if (sy.index == -2) {   // 'all': nothing to bind; an object is inferred from the FIRST agent
                if (brings < 0) search(depth + 1);
                else if (n > 0 && c.op_obj[0] >= 0 && bind(symbols[brings], c.op_obj[0])) search(depth + 1);
            } else if (sy.index < n) {     // a fixed agent: the reference only goes on when it can infer an object (RewardEngine.cc:426-438)
                sy.ent_g = sy.group; sy.ent_i = sy.index;
                if (brings >= 0 && c.op_obj[sy.index] >= 0 && bind(symbols[brings], c.op_obj[sy.index])) search(depth + 1);
            }
        };
        search(0);
    }
    for (int g = 0; g < NG; g++)
        if (C[g].dirty) HIP_OK(hipMemcpy(groups[g].cur.next_reward, C[g].reward.data(), sizeof(float) * groups[g].n, hipMemcpyHostToDevice));
}

}  // namespace magent_amd
