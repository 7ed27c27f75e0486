/* oracle/main.c — command-line driver of the CPU oracle (TEST INFRASTRUCTURE; also the
 * "port" cpu_baseline that bench.py times).  Usage:
 *   oracle_mc <spec> [p0 p1 ...] [--levels N] [--distinct N] [--dump FILE] [--no-deadlock] [--levels-out]
 *             [--threads T [--max-seconds S]]   (T > 0: the multi-threaded BFS of bfs_mt.c, counts only)
 */
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <spec> [params...] [--levels N] [--distinct N] [--dump FILE]\n", argv[0]); return 2; }
    int64_t p[16]; int np = 0;
    or_options o = {0, 0, 1, 1, NULL};
    int show_levels = 0, threads = 0;
    double max_seconds = 0;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--levels") && i + 1 < argc) o.max_levels = strtoull(argv[++i], 0, 10);
        else if (!strcmp(argv[i], "--distinct") && i + 1 < argc) o.max_distinct = strtoull(argv[++i], 0, 10);
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) o.dump_path = argv[++i];
        else if (!strcmp(argv[i], "--no-deadlock")) o.check_deadlock = 0;
        else if (!strcmp(argv[i], "--levels-out")) show_levels = 1;
        else if (!strcmp(argv[i], "--stop-now")) o.stop_on_violation = 2;
        else if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--max-seconds") && i + 1 < argc) max_seconds = atof(argv[++i]);
        else if (np < 16) p[np++] = strtoll(argv[i], 0, 10);
    }
    static or_result r;
    if (threads > 0 ? oracle_run_mt(argv[1], p, np, &o, threads, max_seconds, &r) : oracle_run(argv[1], p, np, &o, &r)) { fprintf(stderr, "error: %s\n", oracle_last_error()); return 2; }
    static const char *vn[] = {"ok", "invariant", "assert", "deadlock", "spec-error", "budget"};
    printf("{\"spec\": \"%s\", \"verdict\": \"%s\", \"violated_invariant\": %d, \"distinct\": %llu, \"generated\": %llu, "
           "\"queue_left\": %llu, \"depth\": %u, \"trace_len\": %u, \"seconds\": %.3f, \"max_msg_domain\": %llu, "
           "\"max_elections\": %llu, \"max_allLogs\": %llu, \"max_inflight\": %llu, \"max_ser_bytes\": %llu, \"arena_bytes\": %llu, \"threads\": %d}\n",
           argv[1], vn[r.verdict], r.violated_invariant, (unsigned long long)r.distinct, (unsigned long long)r.generated,
           (unsigned long long)r.queue_left, r.depth, r.trace_len, r.seconds, (unsigned long long)r.max_stat[0],
           (unsigned long long)r.max_stat[1], (unsigned long long)r.max_stat[2], (unsigned long long)r.max_stat[3],
           (unsigned long long)r.max_stat[4], (unsigned long long)r.arena_bytes, threads);
    if (show_levels) {
        printf("levels:");
        for (uint32_t l = 0; l < r.depth; l++) printf(" %llu", (unsigned long long)r.level_distinct[l]);
        printf("\n");
    }
    if (r.trace_len) {
        for (uint32_t k = 0; k < r.trace_len; k++)
            printf("State %u: <%s>\n%s\n\n", k + 1, oracle_action_name(argv[1], r.trace_action[k]), oracle_trace_state(k));
    }
    return r.verdict == 0 || r.verdict == 5 ? 0 : 1;
}
