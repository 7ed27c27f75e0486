// tlaeval.h — general TLA+ evaluator on the HOST: what `mc X.tla` falls back to for a TLA+ module that
// has neither a hand lowering nor a PlusCal algorithm the compiler takes (the *Specifying Systems* examples of the reference,
// e.g. examples/SpecifyingSystems/AdvancedExamples/MCInnerSerial.tla, whose TLC log testout2:260-266 is the end-to-end golden).
// It evaluates module TEXT the way TLC does — actions left to right, `x' = e` assigns or tests, `x' \in S` / \E / \/ branch,
// lazily evaluated operator arguments, bounded CHOOSE in a fixed value order, EXTENDS chains and named INSTANCEs, cfg CONSTANTS /
// `<-` overrides / CONSTRAINT / INVARIANT / the safety part of a PROPERTY — and runs TLC's breadth-first search with exact
// de-duplication on whole states.  It is NOT a GPU path and never stands in for one: the lowered specs (atomic_add, pcal_intro,
// MCraft, the snapshot-isolation and Paxos models, compiled PlusCal programs) are refused here, and the report says which
// engine produced it.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace tlaeval {

struct Options {
    uint64_t max_levels = 0, max_distinct = 0;
    bool check_deadlock = true;
    bool symmetry = true;             // false: ignore the cfg's SYMMETRY (TLC run on a copy of the cfg without that line)
    double progress_seconds = 0;  // > 0: print TLC's "Progress(d): ..." lines to stdout at this interval
    std::vector<std::string> search;  // directories searched for EXTENDed / INSTANCEd modules (after the root module's own)
    std::string dump_path;            // every stored state as one line "L<level> /\ v = ... /\ w = ..." (tests compare state SETS)
    std::vector<std::string> dump_order;  // variable order of those lines (empty: declaration order)
};

struct Result {
    uint64_t distinct = 0, generated = 0, queue_left = 0, init_states = 0;
    uint32_t depth = 0;
    size_t n_invariants = 0;     // violated_invariant >= n_invariants: the safety part of PROPERTY number (violated_invariant - n_invariants)
    int verdict = 0;             // MC_V_* of include/tlamc.h
    int violated_invariant = -1; // index into the cfg's INVARIANT list
    std::string violated_name, error_message;
    std::vector<std::string> unchecked_properties;                // cfg PROPERTIES with a liveness part (<>, ~>, WF_ / SF_): NOT checked; the caller must say so
    std::vector<uint64_t> levels;                                 // new distinct states per BFS level
    std::vector<std::pair<std::string, std::string>> trace;       // (action label, state text) of a counterexample
    double seconds = 0;
    // TLC's "No Behavior Spec" mode (a cfg with neither SPECIFICATION nor INIT / NEXT): no states; the ASSUMEs of the module and of what
    // it EXTENDS were evaluated (verdict MC_V_OK, or MC_V_ASSUME with error_message "Assumption line L of module M is false."), and
    // what Print / PrintT printed while they were, one line each, in evaluation order
    bool no_behavior = false;
    uint64_t assumes_checked = 0;
    std::vector<std::string> printed;
};

// `tlc X.tla` on the host.  Returns 0 or a negative MC_E* code with `error` set (parse errors: MC_EPARSE; a module the
// evaluator cannot handle: MC_ENOSPEC).
int check_files(const std::string &tla_path, const std::string &cfg_path, const Options &opt, Result &out, std::string &error);

}  // namespace tlaeval
