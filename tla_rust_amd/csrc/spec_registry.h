// spec_registry.h — the registry of hand-lowered specs: maps an mc_spec_desc to its lowering.
// (The reference selects a spec by file/module name on the `tlc` command line, Makefile:6-7;
// no general TLA+ evaluator in v1, each in-scope spec is hand-lowered.)
#pragma once
#include "../../include/tlamc.h"
#include "spec_paxos.h"
#include "spec_pluscal.h"
#include "spec_raft.h"
#include "spec_ssi.h"
#include "spec_vm.h"

namespace mc {

// capacities of the unordered slot arrays (messages / elections / allLogs); exceeding one
// raises MC_EOVERFLOW, never silently drops
// (defaults 40 / 4 / 16, overridable through params[6..8] of MC_SPEC_RAFT)
using SpecRaft2 = SpecRaft<2>;
using SpecRaft3 = SpecRaft<3>;
using SpecRaft5 = SpecRaft<5>;

template <class F>
int dispatch_spec(const mc_spec_desc *d, F &&f) {
    if (!d || d->nparams > 16) return MC_EBADCFG;
    switch (d->spec_id) {
    case MC_SPEC_ATOMIC_ADD: {
        SpecAtomicAdd::Params p;
        if (SpecAtomicAdd::make_params(d->params, d->nparams, p)) return MC_EBADCFG;
        return f(SpecAtomicAdd{}, p);
    }
    case MC_SPEC_PCAL_INTRO: {
        SpecPcalIntro::Params p;
        if (SpecPcalIntro::make_params(d->params, d->nparams, p)) return MC_EBADCFG;
        return f(SpecPcalIntro{}, p);
    }
    case MC_SPEC_RAFT: {
        RaftParams p;
        if (d->nparams < 5) return MC_EBADCFG;
        switch (d->params[0]) {
        case 2: if (SpecRaft2::make_params(d->params, d->nparams, p)) return MC_EBADCFG; return f(SpecRaft2{}, p);
        case 3: if (SpecRaft3::make_params(d->params, d->nparams, p)) return MC_EBADCFG; return f(SpecRaft3{}, p);
        case 5: if (SpecRaft5::make_params(d->params, d->nparams, p)) return MC_EBADCFG; return f(SpecRaft5{}, p);
        default: return MC_EBADCFG;
        }
    }
    case MC_SPEC_SSI: {
        SsiParams p;
        if (SpecSsi::make_params(d->params, d->nparams, p)) return MC_EBADCFG;
        return f(SpecSsi{}, p);
    }
    case MC_SPEC_PAXOS: {
        PaxosParams p;
        if (SpecPaxos::make_params(d->params, d->nparams, p)) return MC_EBADCFG;
        return f(SpecPaxos{}, p);
    }
    case MC_SPEC_PCAL: {  // compiled PlusCal: params[0] = the mc_program handle (pcal_compile.cpp)
        VmParams p;
        if (vm_make_params(d->params, d->nparams, p)) return MC_EBADCFG;
        if (p.nv <= 16) return f(SpecVm16{}, p);
        if (p.nv <= 32) return f(SpecVm32{}, p);
        if (p.nv <= 64) return f(SpecVm64{}, p);
        return f(SpecVm{}, p);
    }
    default: return MC_EBADCFG;
    }
}

}  // namespace mc
