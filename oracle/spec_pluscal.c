/*
 * oracle/spec_pluscal.c — CPU ORACLE (test infrastructure) for the two root PlusCal specs.
 *
 * The reference ships both specs untranslated (pcal_intro.tla:21 is a placeholder); the
 * translation followed here is the one p-manual.pdf App. B pp.60-64 prescribes, written
 * out in specs/pcal_intro.tla and specs/atomic_add.tla of this repo:
 *   - one action per label, guarded by pc[self] = "<label>";
 *   - `await e` is a plain enabling conjunct (p-manual p.62);
 *   - `assert e` becomes Assert(e, "Failure of assertion at line .., column ...")
 *     (README.md:268-269), evaluated while the successor is generated;
 *   - Next has the extra disjunct (\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars
 *     that prevents a deadlock report on termination (p-manual p.63).
 *
 * atomic_add (reference atomic_add.tla:4-23), generalised to N adders + 1 checker:
 *   Increment(self) (atomic_add.tla:11-15), Check (atomic_add.tla:17-21).
 * pcal_intro (reference pcal_intro.tla:4-23 = committed/atomic variant;
 *   README.md:224-240 = the variant with labels A:/B: whose failing TLC run is the golden
 *   output README.md:267-321).
 */
#include "oracle_int.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ atomic_add */
typedef struct { int n; } aa_ctx;
/* state bytes: [0] = global_counter, [1..n] = pc of adder (0 "Increment", 1 "Done"),
 * [n+1] = pc of checker (0 "Check", 1 "Done") */
static int aa_n_init(void *c) { (void)c; return 1; }
static size_t aa_init(void *c, int k, uint8_t *out) {
    aa_ctx *x = c; (void)k;
    memset(out, 0, x->n + 2);
    return x->n + 2;
}
static void aa_succ(void *c, const uint8_t *s, size_t len, or_emit *em) {
    aa_ctx *x = c;
    uint8_t t[64];
    int n = x->n;
    /* Next == Checker \/ (\E self \in 1..N: AdderProc(self)) \/ termination */
    if (s[n + 1] == 0 && s[0] == n) { /* Check: await global_counter = N */
        memcpy(t, s, len); t[n + 1] = 1;
        em->emit(em, t, len, 1, 0);
    }
    for (int i = 1; i <= n; i++)
        if (s[i] == 0) { /* Increment(self) */
            memcpy(t, s, len); t[0] = s[0] + 1; t[i] = 1;
            em->emit(em, t, len, 0, 0);
        }
    int all_done = 1;
    for (int i = 1; i <= n + 1; i++) all_done &= s[i] == 1;
    if (all_done) em->emit(em, s, len, 2, 0);
}
static size_t aa_print(void *c, const uint8_t *s, size_t len, char *buf, size_t cap) {
    aa_ctx *x = c; (void)len;
    size_t k = 0;
    k += snprintf(buf + k, cap - k, "/\\ global_counter = %d\n/\\ pc = <<", s[0]);
    for (int i = 1; i <= x->n; i++) k += snprintf(buf + k, cap - k, "%s\"%s\"", i > 1 ? ", " : "", s[i] ? "Done" : "Increment");
    k += snprintf(buf + k, cap - k, ", \"%s\">>", s[x->n + 1] ? "Done" : "Check");
    return k;
}
const char *or_atomic_add_action(int a) {
    static const char *nm[] = {"Increment", "Check", "Terminating"};
    return a >= 0 && a < 3 ? nm[a] : "?";
}
int or_spec_atomic_add(const int64_t *p, int np, or_spec *o) {
    if (np < 1 || p[0] < 1 || p[0] > 56) { or_set_error("atomic_add: need 1 <= N <= 56"); return -1; }
    aa_ctx *x = malloc(sizeof *x);
    x->n = (int)p[0];
    o->name = "atomic_add"; o->ctx = x; o->max_state_bytes = 64;
    o->n_init = aa_n_init; o->init = aa_init; o->succ = aa_succ; o->print = aa_print;
    o->action_name = or_atomic_add_action;
    return 0;
}

/* ------------------------------------------------------------------ pcal_intro */
typedef struct { int variant, check_inv, max_money, nproc; } pi_ctx;
enum { PC_TRANSFER = 0, PC_A = 1, PC_B = 2, PC_C = 3, PC_DONE = 4 };
static const char *pi_pcname[] = {"Transfer", "A", "B", "C", "Done"};
/* state bytes: [0] alice (int8), [1] bob (int8), [2] total, [3..3+P) pc, [3+P..3+2P) money */
static int pi_n_init(void *c) {
    pi_ctx *x = c;
    int n = 1;
    for (int i = 0; i < x->nproc; i++) n *= x->max_money;
    return n;
}
static size_t pi_init(void *c, int k, uint8_t *out) {
    pi_ctx *x = c;
    int P = x->nproc;
    out[0] = 10; out[1] = 10; out[2] = 20;  /* pcal_intro.tla:5-6 */
    for (int i = 0; i < P; i++) out[3 + i] = PC_TRANSFER;
    /* money \in [1..P -> 1..MaxMoney] (pcal_intro.tla:9); first process = most significant */
    for (int i = P - 1; i >= 0; i--) { out[3 + P + i] = (uint8_t)(1 + k % x->max_money); k /= x->max_money; }
    return 3 + 2 * P;
}
static void pi_succ(void *c, const uint8_t *s, size_t len, or_emit *em) {
    pi_ctx *x = c;
    int P = x->nproc;
    uint8_t t[32];
    for (int self = 0; self < P; self++) {
        int8_t alice = (int8_t)s[0];
        int money = s[3 + P + self];
        memcpy(t, s, len);
        switch (s[3 + self]) {
        case PC_TRANSFER:
            if (x->variant == 0) { /* pcal_intro.tla:11-15: whole body is one atomic step */
                if (alice >= money) { t[0] = (uint8_t)(alice - money); t[1] = (uint8_t)((int8_t)s[1] + money); }
                t[3 + self] = PC_C;
            } else {               /* README.md:232-236: the test alone, then A, B */
                t[3 + self] = alice >= money ? PC_A : PC_C;
            }
            em->emit(em, t, len, 0, 0);
            break;
        case PC_A:
            t[0] = (uint8_t)(alice - money); t[3 + self] = PC_B;
            em->emit(em, t, len, 1, 0);
            break;
        case PC_B:
            t[1] = (uint8_t)((int8_t)s[1] + money); t[3 + self] = PC_C;
            em->emit(em, t, len, 2, 0);
            break;
        case PC_C: /* pcal_intro.tla:16: assert alice_account >= 0 */
            t[3 + self] = PC_DONE;
            em->emit(em, t, len, 3, alice >= 0 ? 0 : OR_FLAG_ASSERT);
            break;
        default: break;
        }
    }
    int all_done = 1;
    for (int i = 0; i < P; i++) all_done &= s[3 + i] == PC_DONE;
    if (all_done) em->emit(em, s, len, 4, 0);
}
static int pi_inv(void *c, const uint8_t *s, size_t len) {
    pi_ctx *x = c; (void)len;
    if (!x->check_inv) return -1;
    /* MoneyInvariant == alice_account + bob_account = account_total (pcal_intro.tla:23) */
    return ((int8_t)s[0] + (int8_t)s[1] == (int)s[2]) ? -1 : 0;
}
static size_t pi_print(void *c, const uint8_t *s, size_t len, char *buf, size_t cap) {
    pi_ctx *x = c; (void)len;
    int P = x->nproc;
    size_t k = 0;
    k += snprintf(buf + k, cap - k, "/\\ alice_account = %d\n/\\ bob_account = %d\n/\\ account_total = %d\n/\\ pc = <<",
                  (int8_t)s[0], (int8_t)s[1], s[2]);
    for (int i = 0; i < P; i++) k += snprintf(buf + k, cap - k, "%s\"%s\"", i ? ", " : "", pi_pcname[s[3 + i]]);
    k += snprintf(buf + k, cap - k, ">>\n/\\ money = <<");
    for (int i = 0; i < P; i++) k += snprintf(buf + k, cap - k, "%s%d", i ? ", " : "", s[3 + P + i]);
    k += snprintf(buf + k, cap - k, ">>");
    return k;
}
const char *or_pcal_intro_action(int a) {
    static const char *nm[] = {"Transfer", "A", "B", "C", "Terminating"};
    return a >= 0 && a < 5 ? nm[a] : "?";
}
int or_spec_pcal_intro(const int64_t *p, int np, or_spec *o) {
    pi_ctx *x = malloc(sizeof *x);
    x->variant = np > 0 ? (int)p[0] : 0;
    x->check_inv = np > 1 ? (int)p[1] : 1;
    x->max_money = np > 2 ? (int)p[2] : 20;
    x->nproc = np > 3 ? (int)p[3] : 2;
    if (x->nproc < 1 || x->nproc > 6 || x->max_money < 1 || x->max_money > 40) {
        or_set_error("pcal_intro: bad parameters"); free(x); return -1;
    }
    o->name = "pcal_intro"; o->ctx = x; o->max_state_bytes = 32;
    o->n_init = pi_n_init; o->init = pi_init; o->succ = pi_succ; o->invariant = pi_inv; o->print = pi_print;
    o->action_name = or_pcal_intro_action;
    return 0;
}
