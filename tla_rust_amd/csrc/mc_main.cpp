// mc — command-line front door of libtlamc.so; the drop-in for `tlc X.tla` of the reference's
// Makefile:6-7 / README.md:262 (same inputs: X.tla with X.cfg beside it; same report lines:
// README.md:267-321).  All work happens behind the C ABI (include/tlamc.h).
//
//   mc X.tla [-config X.cfg] [-deadlock] [-workers N] [-device D] [-generic] [-dump FILE]
//            [-maxdistinct N] [-maxlevels N] [-tablelog2 T] [-arena N] [-chunk N]
//            [-checkpoint FILE] [-recover FILE] [-gpus P [-samedevice | -torch] [-exchange exact|measured|packed] [-fanout N]] [-noprogress] [-I DIR]
//   mc --transpile X.tla [Y.tla ...]      the `pcal2tla *tla` of the reference's Makefile:3-4: inserts (or
//                                         replaces) the TLA+ translation of the PlusCal algorithm in place,
//                                         the previous text is kept as X.old
//
// A module that has a GPU lowering (the PlusCal programs, raft, the snapshot-isolation and Paxos models) is checked on the GPU
// and nowhere else.  Any OTHER TLA+ module (the Specifying Systems examples of the reference: `mc MCInnerSerial.tla` prints
// testout2:260-266) is evaluated on the host by the general evaluator (csrc/tlaeval.cpp); the report's first line says so.
// -I DIR    : one more directory searched for the modules X.tla EXTENDS / INSTANCEs (after X.tla's own and $TLA_PATH's, which
//             is a ':'-separated list).
// -generic  : check a PlusCal module through the compiled program even when a hand lowering exists.
// -jit      : a compiled PlusCal program runs as GENERATED code — translated to straight-line C++ and built for the device (hipcc) when
//             the engine is created, cached by the program's hash — instead of being interpreted on the device (MC_F_JIT; same report).
//             Without it the search starts on the interpreter and moves to the generated code by itself when it lasts long enough for
//             the build to finish first ($TLAMC_AUTOJIT_AFTER seconds before the build starts, default 0.3; $TLAMC_AUTOJIT=0: never).
// -unverified: an MC wrapper (specs/MCraft.tla ...) EXTENDS a module of the reference (raft.tla); when that module is found
//             neither beside the wrapper nor under $TLA_PATH the run is refused, unless this option accepts the built-in
//             lowering unchecked (the report then starts with a warning).
// -dump FILE: like TLC's -dump, write every distinct state found to FILE.
// -checkpoint FILE / -recover FILE: TLC's checkpointing (testout1:10) and -recover: write the run (all states found, level
//             boundaries, counters, parent pointers) after a search that stopped on -maxlevels / -maxdistinct without an
//             error; continue such a run later, with the same X.tla / X.cfg.  With -gpus P: one file per rank,
//             FILE.rank<r>of<P> (the rank's arena, parent pointers and seen-set slice); -recover needs the same P.
// -gpus P   : the search sharded over P GPUs of this node, one process per GPU, collectives over RCCL (include/tlamc.h
//             mc_comm_* / mc_shard_run): mc starts P copies of itself (rank r on device r; rank 0 writes the communicator id to
//             a temporary file the others read) and rank 0 prints TLC's counter / depth lines and, on an error, the behaviour that
//             leads to it — walked back across the ranks' arenas (mc_shard_trace); exit status as below.  A rank that fails takes
//             the others down with it (no rank is left waiting in a collective).  -samedevice: every rank on device D (a one-GPU
//             box: only with a librccl stand-in named by $TLAMC_RCCL — RCCL itself refuses two ranks on one device).
// -gpus P -torch: the same search driven by tla_rust_amd/mc_multi.py over torch.distributed instead — mc replaces itself by
//             `python3 -m torch.distributed.run --nproc-per-node P -m tla_rust_amd.mc_multi X.tla <the other options>`; that front
//             door also rebalances drifting ranks by moving states and prints a counterexample walked back across the ranks
//             ($PYTHON names another interpreter, $MASTER_PORT the rendezvous port).
// -deadlock : as with TLC, do NOT check for deadlock.  -workers is accepted and ignored (the
// GPU is the worker pool).  Exit status: 0 no error, 12 safety violation (invariant / assert),
// 11 deadlock, 1 any other failure — TLC's convention.
#include <limits.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "../../include/tlamc.h"

static int transpile(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "mc: cannot read %s\n", path); return 1; }
    std::string text;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    const int need = mc_pcal_translate(text.c_str(), nullptr, 0);
    if (need < 0) { fprintf(stderr, "mc: %s: %s\n", path, mc_last_error()); return 1; }
    std::vector<char> out((size_t)need + 1);
    mc_pcal_translate(text.c_str(), out.data(), out.size());
    std::string old = path;
    const size_t dot = old.rfind(".tla");
    if (dot != std::string::npos) old.replace(dot, 4, ".old"); else old += ".old";
    if ((f = fopen(old.c_str(), "wb"))) { fwrite(text.data(), 1, text.size(), f); fclose(f); }
    if (!(f = fopen(path, "wb"))) { fprintf(stderr, "mc: cannot write %s\n", path); return 1; }
    fwrite(out.data(), 1, (size_t)need, f);
    fclose(f);
    printf("pcal2tla-compatible translation written to %s (previous text in %s)\n", path, old.c_str());
    return 0;
}

// mc X.tla -gpus P ...: hand over to the multi-process front door.  The package root is two directories above this
// binary (tla_rust_amd/_build/mc).
static int exec_multi(int gpus, int argc, char **argv) {
    char exe[PATH_MAX];
    const ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
    if (n <= 0) { fprintf(stderr, "mc: cannot locate the package root for -gpus\n"); return 1; }
    exe[n] = 0;
    std::string root(exe);
    for (int up = 0; up < 3; up++) { const size_t s = root.rfind('/'); if (s == std::string::npos) break; root.resize(s); }
    const char *pp = getenv("PYTHONPATH");
    setenv("PYTHONPATH", pp && *pp ? (root + ":" + pp).c_str() : root.c_str(), 1);
    const char *port = getenv("MASTER_PORT");
    const char *py = getenv("PYTHON");  // the interpreter that has torch; default: python3 on PATH
    std::vector<std::string> a = {py && *py ? py : "python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=" + std::to_string(gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", port && *port ? port : "29517", "-m", "tla_rust_amd.mc_multi"};
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-gpus")) { ++i; continue; }
        if (!strcmp(argv[i], "-torch") || !strcmp(argv[i], "-samedevice")) continue;
        if (!strcmp(argv[i], "-deadlock") || !strcmp(argv[i], "-dump") || !strcmp(argv[i], "-checkpoint") || !strcmp(argv[i], "-recover")) {
            fprintf(stderr, "mc: %s is not available with -gpus\n", argv[i]);
            return 1;
        }
        a.push_back(argv[i]);
    }
    std::vector<char *> v;
    for (auto &x : a) v.push_back(&x[0]);
    v.push_back(nullptr);
    execvp(v[0], v.data());
    perror("mc: cannot start python3");
    return 1;
}

// ---- mc X.tla -gpus P, native: P copies of this binary, one per GPU, over the hip-rccl back-end of the C ABI
static double now_s() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static int spawn_ranks(int gpus, char **argv) {
    char idfile[] = "/tmp/mc_comm_id_XXXXXX";
    const int fd = mkstemp(idfile);
    if (fd < 0) { perror("mc: mkstemp"); return 1; }
    close(fd);  // empty: rank 0 fills it (through a rename), the other ranks wait for MC_COMM_ID_BYTES bytes
    setenv("MC_WORLD", std::to_string(gpus).c_str(), 1);
    setenv("MC_IDFILE", idfile, 1);
    std::vector<pid_t> pids;
    for (int r = 0; r < gpus; r++) {
        const pid_t p = fork();
        if (p < 0) { perror("mc: fork"); return 1; }
        if (p == 0) {
            setenv("MC_RANK", std::to_string(r).c_str(), 1);
            execv("/proc/self/exe", argv);
            perror("mc: exec");
            _exit(1);
        }
        pids.push_back(p);
    }
    int worst = 0;
    size_t left = pids.size();
    while (left) {
        int st = 0;
        const pid_t p = waitpid(-1, &st, 0);
        if (p < 0) break;
        bool ours = false;
        for (pid_t &q : pids) if (q == p) { q = -1; ours = true; }
        if (!ours) continue;
        --left;
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 1;
        if (code == 1 || (code > worst && worst != 1)) worst = code;  // 1 (a failure) dominates 11 / 12 (TLC's verdict codes)
        if (code != 0 && code != 11 && code != 12)  // a rank FAILED (a verdict is no failure): the others may be waiting for it in a
            for (pid_t q : pids) if (q > 0) kill(q, SIGTERM);  // collective that will never complete
    }
    unlink(idfile);
    return worst;
}
static uint64_t g_fanout = 0;        // -fanout N: in-model successors per state the buffers of a sharded round allow for (0 = the defaults)
static uint32_t g_shard_flags = 0;  // -exchange exact (default) | measured | packed (mc_shard_opts.flags, include/tlamc.h)
static int run_rank(const char *tla, const char *cfgp, mc_config cfg, int rank, int world, const char *idfile, const char *ckpt, const char *recover) {
    mc_spec_desc desc;
    mc_program *prog = nullptr;
    int rc = mc_resolve_files(tla, cfgp, cfg.flags & (MC_F_GENERIC | MC_F_UNVERIFIED), &desc, &prog);
    if (rc) { fprintf(stderr, "mc[%d]: %s: %s\n", rank, mc_strerror(rc), mc_last_error()); return 1; }
    uint8_t id[MC_COMM_ID_BYTES];
    if (rank == 0) {
        if ((rc = mc_comm_unique_id(id))) { fprintf(stderr, "mc[0]: %s: %s\n", mc_strerror(rc), mc_last_error()); return 1; }
        const std::string tmp = std::string(idfile) + ".w";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { perror("mc: communicator id file"); return 1; }
        fclose(f);
        if (rename(tmp.c_str(), idfile)) { perror("mc: rename"); return 1; }
    } else {
        const double t0 = now_s();
        for (;;) {
            FILE *f = fopen(idfile, "rb");
            const size_t n = f ? fread(id, 1, sizeof id, f) : 0;
            if (f) fclose(f);
            if (n == sizeof id) break;
            if (now_s() - t0 > 120) { fprintf(stderr, "mc[%d]: no communicator id from rank 0\n", rank); return 1; }
            usleep(20000);
        }
    }
    if (!getenv("MC_SAMEDEVICE")) cfg.device += rank;  // rank r on device (-device D) + r
    cfg.shard_rank = (uint32_t)rank;
    cfg.shard_count = (uint32_t)world;
    cfg.flags &= ~MC_F_PROGRESS;  // (MC_F_TRACE stays: a counterexample is walked back across the ranks)
    mc_comm *comm = nullptr;
    if ((rc = mc_comm_create(id, (uint32_t)rank, (uint32_t)world, cfg.device, &comm))) { fprintf(stderr, "mc[%d]: %s: %s\n", rank, mc_strerror(rc), mc_last_error()); return 1; }
    mc_engine *eng = nullptr;
    if ((rc = mc_engine_create(&desc, &cfg, &eng))) { fprintf(stderr, "mc[%d]: %s: %s\n", rank, mc_strerror(rc), mc_last_error()); return 1; }
    mc_shard_opts so;
    memset(&so, 0, sizeof so);
    so.flags = g_shard_flags;  // -exchange
    so.packed_fanout = g_fanout;
    so.move_fanout = g_fanout ? 2 * g_fanout : 0;  // (the default pair is 16 / 32)
    so.chunk_states = cfg.chunk_states;
    so.max_distinct = cfg.max_distinct;
    so.max_levels = cfg.max_levels;
    static mc_shard_stats sstats;
    so.stats = &sstats;
    static mc_result res;
    // -recover FILE / -checkpoint FILE with -gpus P: one file per rank, FILE.rank<r>of<P> (mc_shard_restore / mc_shard_checkpoint)
    auto rank_file = [&](const char *stem) { return std::string(stem) + ".rank" + std::to_string(rank) + "of" + std::to_string(world); };
    if (recover && (rc = mc_shard_restore(eng, rank_file(recover).c_str()))) { fprintf(stderr, "mc[%d]: %s: %s\n", rank, mc_strerror(rc), mc_last_error()); return 1; }
    const double t0 = now_s();
    rc = mc_shard_run(eng, comm, &so, &res);
    const double dt = now_s() - t0;
    if (rc) fprintf(stderr, "mc[%d]: %s: %s\n", rank, mc_strerror(rc), mc_last_error());
    if (sstats.restarts && rank == 0)
        fprintf(stderr, "mc: a level had more successors per state than the exchange buckets allow for; the search was started over %llu time(s), "
                        "each with twice the allowance\n", (unsigned long long)sstats.restarts);
    bool ckpt_done = false;
    if (!rc && ckpt && (res.verdict == MC_V_OK || res.verdict == MC_V_BUDGET)) {
        if ((rc = mc_shard_checkpoint(eng, rank_file(ckpt).c_str()))) fprintf(stderr, "mc[%d]: %s: %s\n", rank, mc_strerror(rc), mc_last_error());
        int64_t mine = rc, every[8] = {0};
        const int grc = mc_comm_all_gather(comm, &mine, every, sizeof mine);  // "completed" only when every rank's file is written
        ckpt_done = !grc;
        for (int p = 0; p < world && ckpt_done; p++) ckpt_done = every[p] == 0;
        if (!ckpt_done && !rc) rc = MC_EBADCFG;
    }
    // the behaviour that leads to an error: walked back parent by parent across the ranks (collective: every rank takes part)
    const size_t W = mc_state_bytes(&desc);
    std::vector<uint8_t> tr_states;
    std::vector<int32_t> tr_slots;
    size_t tr_n = 0;
    int32_t tr_final = -1;
    if (!rc && res.verdict != MC_V_OK && res.verdict != MC_V_BUDGET && (cfg.flags & MC_F_TRACE)) {
        tr_n = 4096;
        tr_states.resize(tr_n * W);
        tr_slots.resize(tr_n);
        const int trc = mc_shard_trace(eng, comm, tr_states.data(), tr_slots.data(), &tr_n, &tr_final);
        if (trc) { fprintf(stderr, "mc[%d]: counterexample: %s: %s\n", rank, mc_strerror(trc), mc_last_error()); tr_n = 0; }
    }
    if (!rc && rank == 0) {  // TLC's closing lines (README.md:319-320, testout2:260-266), as tla_rust_amd/mc_multi.py prints them
        const unsigned long long n0 = res.levels ? (unsigned long long)res.level_distinct[0] : 0ull;
        printf("Finished computing initial states: %llu distinct state%s generated.\n", n0, n0 == 1 ? "" : "s");
        if (res.verdict == MC_V_OK) printf("Model checking completed. No error has been found.\n");
        else if (res.verdict == MC_V_BUDGET) printf("Search stopped by the level/state budget; no error has been found so far.\n");
        if (ckpt_done) printf("-- Checkpointing of run %s completed.\n", ckpt);  // testout1:10 (one file per rank: %s.rank<r>of<P>)
        if (res.verdict != MC_V_OK && res.verdict != MC_V_BUDGET) {
            printf("%s\n", res.verdict == MC_V_INVARIANT ? "Error: Invariant is violated." : res.verdict == MC_V_ASSERT ? "Error: The first argument of Assert evaluated to FALSE."
                          : res.verdict == MC_V_DEADLOCK ? "Error: Deadlock reached." : "Error: TLC would raise an evaluation error.");
            if (tr_n) {  // README.md:270-311: "State k: <Action>" + the variables
                printf("Error: The behavior up to this point is:\n");
                std::vector<char> text(1 << 16);
                std::vector<uint8_t> succ(W);
                auto print_state = [&](size_t k, const uint8_t *prev, int32_t slot, const uint8_t *st) {
                    const char *name = "Initial predicate";
                    if (prev) { const int a = mc_state_action(&desc, prev, slot); name = a >= 0 ? mc_action_name(&desc, a) : "?"; }
                    const int n = mc_state_format(&desc, st, text.data(), text.size());
                    printf("State %zu: <%s>\n%.*s\n", k + 1, name, n > 0 ? n : 0, text.data());
                };
                for (size_t k = 0; k < tr_n; k++) print_state(k, k ? &tr_states[(k - 1) * W] : nullptr, tr_slots[k], &tr_states[k * W]);
                if (tr_final >= 0 && mc_state_apply(&desc, &tr_states[(tr_n - 1) * W], tr_final, succ.data()) == 0)
                    print_state(tr_n, &tr_states[(tr_n - 1) * W], tr_final, succ.data());
            } else {
                printf("(no behavior: the engines keep no parent pointers)\n");
            }
        }
        printf("%llu states generated, %llu distinct states found, %llu states left on queue.\n", (unsigned long long)res.generated,
               (unsigned long long)res.distinct, (unsigned long long)res.queue_left);
        printf("The depth of the complete state graph search is %u.\n", res.depth);
        printf("(%d GPU%s over RCCL, %.3f s, %.3g distinct states/s)\n", world, world == 1 ? "" : "s", dt, (double)res.distinct / (dt > 1e-9 ? dt : 1e-9));
        if (world > 1 && sstats.routed_candidates)  // rank 0's share of the exchange: what it handed over against 9 bytes per routed candidate
            printf("(exchange, this rank: %.1f MB of fingerprints and answers for %llu candidates = %.2f x the 9 bytes each needs; %llu stay / %llu move levels)\n",
                   (double)sstats.fp_answer_bytes / 1e6, (unsigned long long)sstats.routed_candidates,
                   (double)sstats.fp_answer_bytes / (9.0 * (double)sstats.routed_candidates), (unsigned long long)sstats.stay_levels,
                   (unsigned long long)sstats.move_levels);
        fflush(stdout);
    }
    mc_engine_destroy(eng);
    mc_comm_destroy(comm);
    if (prog) mc_program_free(prog);
    if (rc) return 1;
    if (res.verdict == MC_V_OK || res.verdict == MC_V_BUDGET) return 0;
    return res.verdict == MC_V_DEADLOCK ? 11 : 12;
}

int main(int argc, char **argv) {
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);   // (RCCL between processes needs dmabuf IPC on this driver; read when the HSA runtime starts)
    int gpus = 0;
    bool torch_door = false;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-gpus") && i + 1 < argc) {
            gpus = atoi(argv[i + 1]);
            if (gpus < 1 || gpus > 64) { fprintf(stderr, "mc: -gpus needs a number of GPUs\n"); return 1; }
        }
        if (!strcmp(argv[i], "-torch")) torch_door = true;
        if (!strcmp(argv[i], "-samedevice")) setenv("MC_SAMEDEVICE", "1", 1);
    }
    if (gpus && torch_door) return exec_multi(gpus, argc, argv);
    const char *env_rank = getenv("MC_RANK");
    if (gpus && !env_rank) {
        if (gpus > 8) { fprintf(stderr, "mc: -gpus: at most 8 ranks (one node)\n"); return 1; }
        return spawn_ranks(gpus, argv);
    }
    if (argc >= 2 && (!strcmp(argv[1], "--transpile") || !strcmp(argv[1], "-transpile"))) {
        int rc = argc > 2 ? 0 : 1;
        for (int i = 2; i < argc; i++) rc |= transpile(argv[i]);
        return rc;
    }
    const char *tla = nullptr, *cfgp = nullptr, *dump = nullptr, *recover = nullptr, *ckpt = nullptr;
    mc_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.flags = MC_F_DEADLOCK | MC_F_TRACE | MC_F_PROGRESS;  // TLC reports its progress while it runs (testout2:4-259); -noprogress
    cfg.table_capacity = 1ull << 26;
    cfg.arena_capacity = 1ull << 24;
    for (int i = 1; i < argc; i++) {
        auto arg = [&](const char *name) { return !strcmp(argv[i], name) && i + 1 < argc; };
        if (arg("-config")) cfgp = argv[++i];
        else if (!strcmp(argv[i], "-deadlock")) cfg.flags &= ~MC_F_DEADLOCK;
        else if (arg("-dump")) dump = argv[++i];
        else if (arg("-checkpoint")) ckpt = argv[++i];
        else if (arg("-recover")) recover = argv[++i];
        else if (arg("-workers")) ++i;
        else if (arg("-gpus")) ++i;
        else if (arg("-fanout")) {  // a dense model (the SI specs: up to ~40 in-model successors per state) without the restarts that find the allowance by doubling
            g_fanout = strtoull(argv[++i], nullptr, 10);
            if (g_fanout < 1 || g_fanout > 4096) { fprintf(stderr, "mc: -fanout needs a number of successors per state (1..4096)\n"); return 1; }
        }
        else if (arg("-exchange")) {  // how the stay levels of a -gpus run exchange their candidates (MC_SHARD_*)
            const char *v = argv[++i];
            if (!strcmp(v, "packed")) g_shard_flags = MC_SHARD_PACKED | MC_SHARD_FIXED_CAPS;
            else if (!strcmp(v, "measured")) g_shard_flags = MC_SHARD_PACKED;
            else if (!strcmp(v, "exact")) g_shard_flags = 0;
            else { fprintf(stderr, "mc: -exchange exact | measured | packed\n"); return 1; }
        }
        else if (!strcmp(argv[i], "-torch") || !strcmp(argv[i], "-samedevice")) {}
        else if (!strcmp(argv[i], "-generic")) cfg.flags |= MC_F_GENERIC;
        else if (!strcmp(argv[i], "-jit")) cfg.flags |= MC_F_JIT;
        else if (!strcmp(argv[i], "-unverified")) cfg.flags |= MC_F_UNVERIFIED;
        else if (arg("-I")) {  // one more directory searched for EXTENDed / INSTANCEd modules (appended to $TLA_PATH)
            const char *old = getenv("TLA_PATH");
            const std::string v = old && *old ? std::string(old) + ":" + argv[i + 1] : std::string(argv[i + 1]);
            setenv("TLA_PATH", v.c_str(), 1);
            ++i;
        }
        else if (!strcmp(argv[i], "-noprogress")) cfg.flags &= ~MC_F_PROGRESS;
        else if (arg("-device")) cfg.device = atoi(argv[++i]);
        else if (arg("-maxdistinct")) cfg.max_distinct = strtoull(argv[++i], 0, 10);
        else if (arg("-maxlevels")) cfg.max_levels = strtoull(argv[++i], 0, 10);
        else if (arg("-tablelog2")) cfg.table_capacity = 1ull << atoi(argv[++i]);
        else if (arg("-table")) cfg.table_capacity = strtoull(argv[++i], 0, 10);  // seen-set slots, any number
        else if (arg("-arena")) cfg.arena_capacity = strtoull(argv[++i], 0, 10);
        else if (arg("-chunk")) cfg.chunk_states = strtoull(argv[++i], 0, 10);
        else if (!strcmp(argv[i], "-help") || !strcmp(argv[i], "--help") || !strcmp(argv[i], "-h")) { tla = nullptr; break; }
        else if (argv[i][0] != '-') tla = argv[i];
        else { fprintf(stderr, "mc: unknown option %s\n", argv[i]); return 1; }
    }
    if (!tla) {
        fprintf(stderr,
                "usage: mc X.tla [-config X.cfg] [-deadlock] [-dump FILE] [-generic] [-jit] [-unverified] [-device D] [-I DIR]\n"
                "                [-maxdistinct N] [-maxlevels N] [-tablelog2 T] [-arena N] [-chunk N]\n"
                "                [-checkpoint FILE] [-recover FILE] [-gpus P [-torch]]                    check X.tla like `tlc X.tla`\n"
                "       mc --transpile X.tla [Y.tla ...]                                                  translate like `pcal2tla`\n"
                "exit status: 0 no error, 12 invariant / assertion violated, 11 deadlock, 1 anything else\n");
        return 1;
    }
    if (gpus) {  // one rank of `mc X.tla -gpus P`
        if (dump) { fprintf(stderr, "mc: -dump is not available with -gpus\n"); return 1; }
        return run_rank(tla, cfgp, cfg, atoi(env_rank), gpus, getenv("MC_IDFILE"), ckpt, recover);
    }
    std::vector<char> report(1 << 22);
    static mc_result res;
    const int rc = mc_check_files_ckpt(tla, cfgp, &cfg, report.data(), report.size(), &res, dump, recover, ckpt);
    if (rc) {
        fprintf(stderr, "mc: %s: %s\n", mc_strerror(rc), mc_last_error());
        return 1;
    }
    fputs(report.data(), stdout);
    if (res.verdict == MC_V_OK || res.verdict == MC_V_BUDGET) return 0;
    return res.verdict == MC_V_DEADLOCK ? 11 : 12;
}
