/*
 * solve_qp_json.c -- qp_in.json [init.json] -> tdunes on the MI355X -> qp_out.json on stdout.
 *
 * The wire format and the flow are those of the reference's JSON front end
 * (examples/solve_qp_json.cpp:206-612, written there in C++ on nlohmann/json + boost; both absent
 * here, so this is plain C with its own small JSON reader):
 *   input   "nodes": [{Q, R, S, q, r, lx, lu, ux, uu}], "edges": [{from, to, A, B, b}],
 *           optional "options": {solver, maxit, stationarityTolerance, lineSearchMaxIter,
 *           lineSearchBeta, lineSearchGamma, checkLastActiveSet, clipping, regType, regTol, regValue}
 *   init    optional second file {x0, lam0_tree}: pins x0, warm-starts the duals     (:212-217,311-325,410-413)
 *   output  {"init": {"lam0_tree"}, "solution": {"nodes": [{x, mu_x, u, mu_u, mu_d}], "edges": [{lam}]},
 *            "info": {solver, cpu_time, status, num_iter, kkt_tol}}                  (:127-171,568-580)
 * Matrices are arrays of rows (a 1 x n or n x 1 matrix may be a flat array, a 1 x 1 matrix a number),
 * as readColMajorMatrix / readVector accept them (:73-110).
 *
 * Differences, all on the permissive side:
 *   - only "solver": "tdunes" exists in this build (sdunes / hpmpc are out of scope): anything else exits with -1;
 *   - lx/lu/ux/uu may be missing (then -1e12 / +1e12), so that the reference's unit-test fixtures
 *     examples/random_qp_utils/data0x.json, which carry no bounds, can be replayed;
 *   - x0 is eliminated (tree_qp_in_eliminate_x0, :353) only when node 0 is pinned by equal bounds;
 *   - `--dims` prints the parsed dimensions and exits without touching a device (used by the CPU tests).
 */
#include <assert.h>
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "treeqp/src/tree_qp_common.h"
#include "treeqp/src/dual_Newton_tree.h"
#include "treeqp/utils/types.h"
#include "treeqp/utils/tree.h"

#ifndef NREP
#define NREP 1
#endif

/* ------------------------------------------------------------------------------------------ */
/* a small JSON reader: values live in one arena, objects and arrays are linked lists        */
/* ------------------------------------------------------------------------------------------ */

typedef enum { J_NULL, J_BOOL, J_NUM, J_STR, J_ARR, J_OBJ } jkind;

typedef struct jval {
    jkind kind;
    double num;              /* J_NUM, J_BOOL */
    char *str;               /* J_STR; key when the value is a member of an object */
    char *key;
    struct jval *child;      /* first element / member */
    struct jval *next;       /* sibling */
    int count;               /* elements / members */
} jval;

typedef struct { const char *p, *end; const char *err; } jreader;

static void jskip(jreader *r) { while (r->p < r->end && isspace((unsigned char)*r->p)) r->p++; }

static jval *jnew(jkind k) { jval *v = calloc(1, sizeof(jval)); if (!v) { perror("calloc"); exit(1); } v->kind = k; return v; }

static char *jstring(jreader *r)
{
    if (r->p >= r->end || *r->p != '"') { r->err = "expected a string"; return NULL; }
    r->p++;
    size_t cap = 32, n = 0;
    char *s = malloc(cap);
    while (r->p < r->end && *r->p != '"') {
        char c = *r->p++;
        if (c == '\\' && r->p < r->end) {
            char e = *r->p++;
            switch (e) {
                case 'n': c = '\n'; break; case 't': c = '\t'; break; case 'r': c = '\r'; break;
                case 'b': c = '\b'; break; case 'f': c = '\f'; break;
                case 'u': r->p += (r->end - r->p >= 4) ? 4 : 0; c = '?'; break;      /* names here are ASCII */
                default: c = e;
            }
        }
        if (n + 2 > cap) { cap *= 2; s = realloc(s, cap); }
        s[n++] = c;
    }
    if (r->p >= r->end) { r->err = "unterminated string"; free(s); return NULL; }
    r->p++;
    s[n] = 0;
    return s;
}

static jval *jparse(jreader *r)
{
    jskip(r);
    if (r->p >= r->end) { r->err = "unexpected end of input"; return NULL; }
    const char c = *r->p;
    if (c == '{' || c == '[') {
        const int is_obj = c == '{';
        jval *v = jnew(is_obj ? J_OBJ : J_ARR), **tail = &v->child;
        r->p++;
        jskip(r);
        if (r->p < r->end && *r->p == (is_obj ? '}' : ']')) { r->p++; return v; }
        for (;;) {
            char *key = NULL;
            jskip(r);
            if (is_obj) {
                key = jstring(r);
                if (!key) return NULL;
                jskip(r);
                if (r->p >= r->end || *r->p != ':') { r->err = "expected ':'"; return NULL; }
                r->p++;
            }
            jval *e = jparse(r);
            if (!e) return NULL;
            e->key = key;
            *tail = e; tail = &e->next; v->count++;
            jskip(r);
            if (r->p < r->end && *r->p == ',') { r->p++; continue; }
            if (r->p < r->end && *r->p == (is_obj ? '}' : ']')) { r->p++; return v; }
            r->err = "expected ',' or a closing bracket";
            return NULL;
        }
    }
    if (c == '"') { jval *v = jnew(J_STR); v->str = jstring(r); return v->str ? v : NULL; }
    if (!strncmp(r->p, "true", 4)) { r->p += 4; jval *v = jnew(J_BOOL); v->num = 1; return v; }
    if (!strncmp(r->p, "false", 5)) { r->p += 5; jval *v = jnew(J_BOOL); v->num = 0; return v; }
    if (!strncmp(r->p, "null", 4)) { r->p += 4; return jnew(J_NULL); }
    char *endp = NULL;
    const double d = strtod(r->p, &endp);
    if (endp == r->p) { r->err = "unexpected character"; return NULL; }
    r->p = endp;
    jval *v = jnew(J_NUM); v->num = d;
    return v;
}

static const jval *jget(const jval *o, const char *key)
{
    if (!o || o->kind != J_OBJ) return NULL;
    for (const jval *m = o->child; m; m = m->next) if (m->key && !strcmp(m->key, key)) return m;
    return NULL;
}

static void die(const char *what, const char *detail)
{
    fprintf(stderr, "solve_qp_json: %s%s%s\n", what, detail ? ": " : "", detail ? detail : "");
    exit(2);
}

static jval *jload(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) die("cannot open", path);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = malloc((size_t)n + 1);
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) die("cannot read", path);
    fclose(f);
    buf[n] = 0;
    jreader r = { buf, buf + n, NULL };
    jval *v = jparse(&r);
    if (!v) die(r.err ? r.err : "parse error", path);
    return v;                                  /* buf is referenced by nothing: strings were copied */
}

/* size of a vector-like value: a number counts as one entry, null / a missing value as none */
static int jlen(const jval *v) { return !v || v->kind == J_NULL ? 0 : (v->kind == J_ARR ? v->count : 1); }

static double jnum(const jval *v, const char *what)
{
    if (!v || (v->kind != J_NUM && v->kind != J_BOOL)) die("expected a number in", what);
    return v->num;
}

/* readVector (:73-88) */
static void read_vector(const jval *v, int n, double *out, const char *what)
{
    if (n == 0) return;
    if (!v) die("missing vector", what);
    if (v->kind != J_ARR) { if (n != 1) die("vector has the wrong length", what); out[0] = jnum(v, what); return; }
    if (v->count != n) die("vector has the wrong length", what);
    int i = 0;
    for (const jval *e = v->child; e; e = e->next) {
        /* a column given as [[a],[b],...] */
        out[i++] = (e->kind == J_ARR && e->count == 1) ? jnum(e->child, what) : jnum(e, what);
    }
}

/* readColMajorMatrix (:92-110): M x N, array of rows */
static void read_matrix(const jval *v, int M, int N, double *out, const char *what)
{
    if (M == 0 || N == 0) return;
    if (!v) die("missing matrix", what);
    if (M == 1 && !(v->kind == J_ARR && v->count == 1 && v->child->kind == J_ARR)) { read_vector(v, N, out, what); return; }
    if (N == 1 && !(v->kind == J_ARR && v->child && v->child->kind == J_ARR)) { read_vector(v, M, out, what); return; }
    if (v->kind != J_ARR || v->count != M) die("matrix has the wrong number of rows", what);
    int i = 0;
    for (const jval *row = v->child; row; row = row->next, i++) {
        if (row->kind != J_ARR || row->count != N) die("matrix has the wrong number of columns", what);
        int j = 0;
        for (const jval *e = row->child; e; e = e->next, j++) out[i + (size_t)j * M] = jnum(e, what);
    }
}

/* convert_reg_type (:59-69) */
static regType_t reg_type_of(const char *s)
{
    if (!strcmp(s, "TREEQP_NO_REGULARIZATION")) return TREEQP_NO_REGULARIZATION;
    if (!strcmp(s, "TREEQP_ALWAYS_LEVENBERG_MARQUARDT")) return TREEQP_ALWAYS_LEVENBERG_MARQUARDT;
    if (!strcmp(s, "TREEQP_ON_THE_FLY_LEVENBERG_MARQUARDT")) return TREEQP_ON_THE_FLY_LEVENBERG_MARQUARDT;
    return TREEQP_UNKNOWN_REGULARIZATION;
}

static void put_vec(FILE *o, const double *v, int n)
{
    fputc('[', o);
    for (int i = 0; i < n; i++) fprintf(o, "%s%.17g", i ? ", " : "", v[i]);
    fputc(']', o);
}

int main(int argc, char **argv)
{
    int dims_only = 0, nfiles = 0;
    const char *files[2] = { NULL, NULL };
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--dims")) dims_only = 1;
        else if (nfiles < 2) files[nfiles++] = argv[i];
    }
    if (nfiles < 1) die("no input files", "usage: treeqp_solve_json [--dims] qp_in.json [init.json] > qp_out.json");

    const jval *j_in = jload(files[0]);
    const jval *j_init = nfiles > 1 ? jload(files[1]) : NULL;
    const jval *nodes = jget(j_in, "nodes"), *edges = jget(j_in, "edges");
    if (!nodes || nodes->kind != J_ARR || !edges || edges->kind != J_ARR) die("need \"nodes\" and \"edges\" arrays", files[0]);
    const int Nn = nodes->count;
    if (edges->count != Nn - 1) die("a tree with N nodes has N - 1 edges", files[0]);

    /* dimensions (:231-248) */
    int *nx = calloc((size_t)Nn, sizeof(int)), *nu = calloc((size_t)Nn, sizeof(int)), *nc = calloc((size_t)Nn, sizeof(int)), *nk = calloc((size_t)Nn, sizeof(int));
    int maxd = 1;
    {
        int i = 0;
        for (const jval *n = nodes->child; n; n = n->next, i++) {
            nx[i] = jlen(jget(n, "q")); nu[i] = jlen(jget(n, "r")); nc[i] = jlen(jget(n, "ld"));
            if (nx[i] > maxd) maxd = nx[i];
            if (nu[i] > maxd) maxd = nu[i];
        }
        for (const jval *e = edges->child; e; e = e->next) {
            const int from = (int)jnum(jget(e, "from"), "edge.from");
            if (from < 0 || from >= Nn) die("edge.from out of range", files[0]);
            nk[from]++;
        }
    }
    if (dims_only) {
        printf("{\"Nn\": %d, \"nx\": [", Nn);
        for (int i = 0; i < Nn; i++) printf("%s%d", i ? ", " : "", nx[i]);
        printf("], \"nu\": [");
        for (int i = 0; i < Nn; i++) printf("%s%d", i ? ", " : "", nu[i]);
        printf("], \"nk\": [");
        for (int i = 0; i < Nn; i++) printf("%s%d", i ? ", " : "", nk[i]);
        printf("], \"has_options\": %s}\n", jget(j_in, "options") ? "true" : "false");
        return 0;
    }
    for (int i = 0; i < Nn; i++) if (nc[i]) die("general constraints are not part of the tdunes path of this build", files[0]);

    /* QP data (:251-306) */
    tree_qp_in qp_in;
    void *qp_in_memory = malloc((size_t)tree_qp_in_calculate_size(Nn, nx, nu, nc, nk));
    tree_qp_in_create(Nn, nx, nu, nc, nk, &qp_in, qp_in_memory);
    double *M = malloc(sizeof(double) * (size_t)maxd * (size_t)maxd), *v = malloc(sizeof(double) * (size_t)maxd);
    for (const jval *e = edges->child; e; e = e->next) {
        const int to = (int)jnum(jget(e, "to"), "edge.to"), from = (int)jnum(jget(e, "from"), "edge.from");
        if (to < 1 || to >= Nn) die("edge.to out of range", files[0]);
        read_matrix(jget(e, "A"), nx[to], nx[from], M, "edge.A"); tree_qp_in_set_edge_A_colmajor(M, -1, &qp_in, to - 1);
        read_matrix(jget(e, "B"), nx[to], nu[from], M, "edge.B"); tree_qp_in_set_edge_B_colmajor(M, -1, &qp_in, to - 1);
        read_vector(jget(e, "b"), nx[to], v, "edge.b"); tree_qp_in_set_edge_b(v, &qp_in, to - 1);
    }
    {
        int i = 0;
        for (const jval *n = nodes->child; n; n = n->next, i++) {
            read_matrix(jget(n, "Q"), nx[i], nx[i], M, "node.Q"); tree_qp_in_set_node_Q_colmajor(M, -1, &qp_in, i);
            read_matrix(jget(n, "R"), nu[i], nu[i], M, "node.R"); tree_qp_in_set_node_R_colmajor(M, -1, &qp_in, i);
            if (jlen(jget(n, "S"))) read_matrix(jget(n, "S"), nu[i], nx[i], M, "node.S"); else memset(M, 0, sizeof(double) * (size_t)maxd * maxd);
            tree_qp_in_set_node_S_colmajor(M, -1, &qp_in, i);
            read_vector(jget(n, "q"), nx[i], v, "node.q"); tree_qp_in_set_node_q(v, &qp_in, i);
            read_vector(jget(n, "r"), nu[i], v, "node.r"); tree_qp_in_set_node_r(v, &qp_in, i);
#define BOUND(KEY, N, FILL, SETTER)                                                                     \
            if (jget(n, KEY)) read_vector(jget(n, KEY), N, v, "node." KEY); else for (int t = 0; t < (N); t++) v[t] = (FILL); \
            SETTER(v, &qp_in, i)
            BOUND("lx", nx[i], -1e12, tree_qp_in_set_node_xmin);
            BOUND("lu", nu[i], -1e12, tree_qp_in_set_node_umin);
            BOUND("ux", nx[i], 1e12, tree_qp_in_set_node_xmax);
            BOUND("uu", nu[i], 1e12, tree_qp_in_set_node_umax);
#undef BOUND
        }
    }
    if (j_init && nx[0] > 0 && jget(j_init, "x0")) {                     /* :308-321 */
        read_vector(jget(j_init, "x0"), nx[0], v, "init.x0");
        tree_qp_in_set_node_xmin(v, &qp_in, 0);
        tree_qp_in_set_node_xmax(v, &qp_in, 0);
    }

    tree_qp_out qp_out;
    void *qp_out_memory = malloc((size_t)tree_qp_out_calculate_size(Nn, nx, nu, nc));
    tree_qp_out_create(Nn, nx, nu, nc, &qp_out, qp_out_memory);

    const jval *options = jget(j_in, "options");
    const char *solver = "tdunes";
    if (options && jget(options, "solver") && jget(options, "solver")->kind == J_STR) solver = jget(options, "solver")->str;
    if (strcmp(solver, "tdunes")) { fprintf(stderr, "solve_qp_json: solver \"%s\" is out of scope of this build (tdunes only)\n", solver); return -1; }

    /* x0 (:350-353): kept for the output, eliminated when pinned */
    const int nx0 = nx[0];
    double *x0_bkp = malloc(sizeof(double) * (size_t)(nx0 ? nx0 : 1)), *x0_hi = malloc(sizeof(double) * (size_t)(nx0 ? nx0 : 1));
    tree_qp_in_get_node_xmin(x0_bkp, &qp_in, 0);
    tree_qp_in_get_node_xmax(x0_hi, &qp_in, 0);
    int pinned = nx0 > 0;
    for (int t = 0; t < nx0; t++) if (fabs(x0_bkp[t] - x0_hi[t]) > 1e-10) pinned = 0;
    if (pinned) tree_qp_in_eliminate_x0(&qp_in);

    /* options (:357-392) */
    treeqp_tdunes_opts_t opts;
    void *opts_memory = malloc((size_t)treeqp_tdunes_opts_calculate_size(Nn));
    treeqp_tdunes_opts_create(Nn, &opts, opts_memory);
    treeqp_tdunes_opts_set_default(Nn, &opts);
    for (int i = 0; i < Nn; i++) opts.qp_solver[i] = TREEQP_QPOASES_SOLVER;
    if (options) {
#define OPT_NUM(KEY, FIELD, TYPE) if (jget(options, KEY)) opts.FIELD = (TYPE)jnum(jget(options, KEY), "options." KEY)
        OPT_NUM("maxit", maxIter, int);
        OPT_NUM("stationarityTolerance", stationarityTolerance, double);
        OPT_NUM("lineSearchMaxIter", lineSearchMaxIter, int);
        OPT_NUM("lineSearchBeta", lineSearchBeta, double);
        OPT_NUM("lineSearchGamma", lineSearchGamma, double);
        OPT_NUM("checkLastActiveSet", checkLastActiveSet, int);
        OPT_NUM("regTol", regTol, double);
        OPT_NUM("regValue", regValue, double);
#undef OPT_NUM
        const jval *clip = jget(options, "clipping");
        if (clip && clip->num != 0.0) for (int i = 0; i < Nn; i++) opts.qp_solver[i] = TREEQP_CLIPPING_SOLVER;
        const jval *rt = jget(options, "regType");
        if (rt && rt->kind == J_STR) opts.regType = reg_type_of(rt->str);
    }

    treeqp_tdunes_workspace work;
    void *solver_memory = malloc((size_t)treeqp_tdunes_calculate_size(&qp_in, &opts));
    treeqp_tdunes_create(&qp_in, &opts, &work, solver_memory);

    const int dim_lam = total_number_of_dynamic_constraints(&qp_in);
    double *lam0_tree = calloc((size_t)(dim_lam ? dim_lam : 1), sizeof(double));
    if (j_init && jget(j_init, "lam0_tree")) read_vector(jget(j_init, "lam0_tree"), dim_lam, lam0_tree, "init.lam0_tree");

    int status = 0, prev_status = 0, num_iter = 0;
    double min_time = 0.0;
    for (int rep = 0; rep < NREP; rep++) {                                /* :416-432 */
        treeqp_tdunes_set_dual_initialization(lam0_tree, &work);
        status = (int)treeqp_tdunes_solve(&qp_in, &qp_out, &opts, &work);
        if (rep == 0) { min_time = qp_out.info.total_time; num_iter = qp_out.info.iter; }
        else {
            if (qp_out.info.total_time < min_time) min_time = qp_out.info.total_time;
            assert(status == prev_status);
            assert(num_iter == qp_out.info.iter);
        }
        prev_status = status;
    }
    {   /* tdunes_update_multipliers (:193-203) */
        int idx = 0;
        for (int e = 0; e < Nn - 1; e++) { tree_qp_out_get_edge_lam(&lam0_tree[idx], &qp_out, e); idx += qp_in.nx[e + 1]; }
    }
    const double kkt_err = tree_qp_out_max_KKT_res(&qp_in, &qp_out);

    /* output (:127-171, 566-580) */
    FILE *o = stdout;
    fprintf(o, "{\n  \"init\": {\"lam0_tree\": "); put_vec(o, lam0_tree, dim_lam);
    fprintf(o, "},\n  \"solution\": {\n    \"nodes\": [\n");
    for (int i = 0; i < Nn; i++) {
        const int nxi = (i == 0 && pinned) ? 0 : nx[i];
        fprintf(o, "      {\"x\": ");
        if (i == 0 && pinned) put_vec(o, x0_bkp, nx0);                    /* :570 */
        else { tree_qp_out_get_node_x(v, &qp_out, i); put_vec(o, v, nxi); }
        fprintf(o, ", \"mu_x\": "); if (nxi) tree_qp_out_get_node_mu_x(v, &qp_out, i); put_vec(o, v, nxi);
        fprintf(o, ", \"u\": "); if (nu[i]) tree_qp_out_get_node_u(v, &qp_out, i); put_vec(o, v, nu[i]);
        fprintf(o, ", \"mu_u\": "); if (nu[i]) tree_qp_out_get_node_mu_u(v, &qp_out, i); put_vec(o, v, nu[i]);
        fprintf(o, ", \"mu_d\": []}%s\n", i + 1 < Nn ? "," : "");
    }
    fprintf(o, "    ],\n    \"edges\": [\n");
    for (int e = 0; e < Nn - 1; e++) {
        tree_qp_out_get_edge_lam(v, &qp_out, e);
        fprintf(o, "      {\"lam\": "); put_vec(o, v, qp_in.nx[e + 1]); fprintf(o, "}%s\n", e + 2 < Nn ? "," : "");
    }
    fprintf(o, "    ]\n  },\n  \"info\": {\"solver\": \"%s\", \"cpu_time\": %.9g, \"status\": %d, \"num_iter\": %d, \"kkt_tol\": %.17g}\n}\n",
            solver, min_time, status, qp_out.info.iter, kkt_err);

    treeqp_tdunes_destroy(&work);
    free(solver_memory); free(opts_memory); free(qp_out_memory); free(qp_in_memory);
    return 0;
}
