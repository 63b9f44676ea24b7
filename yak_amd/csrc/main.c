/*
 * main.c -- `yak-amd`: the repo's own small command-line driver of libyak_amd.so, plain C against
 * include/yak.h only.  It is a test and benchmark vehicle (tests/, bench.py's e2e_cli figure), not a
 * re-creation of the reference's CLI: that one runs unmodified on the library (INTEGRATION.md section 2,
 * oracle/_ref/yak_on_amd).  Two sub-commands drive the two call sequences the library serves:
 *     count   the counting protocol behind reference main.c:53-61 (one pass, or two passes + shrink
 *             when a bloom filter is asked for)
 *     qv      the lookup protocol behind reference main.c:163-215 (restore, histogram, yak_qv, solve)
 * Option letters follow the reference so that test command lines can be shared; the parser, the
 * sub-command table and the usage texts are this file's own.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "yak.h"
void yakamd_test_set(const char *name, int64_t value);        /* include/yak_amd.h: the one entry point beyond yak.h this driver knows, for -X */

/* ---- a table-driven option scanner: "-x", "-xVALUE" and "-x VALUE"; stops at the first non-option ---- */
enum arg_kind { ARG_FLAG, ARG_I32, ARG_SIZE, ARG_I64SIZE, ARG_F64, ARG_TEXT };
struct arg_def { char letter; enum arg_kind kind; void *dst; const char *what; };

static double with_suffix(const char *s)                     /* 64m, 3.2g, 100k: decimal multipliers */
{
	char *end = 0;
	double v = strtod(s, &end);
	switch (end ? *end : 0) {
	case 'k': case 'K': return v * 1e3;
	case 'm': case 'M': return v * 1e6;
	case 'g': case 'G': return v * 1e9;
	default: return v;
	}
}

static void arg_store(const struct arg_def *d, const char *text)
{
	switch (d->kind) {
	case ARG_FLAG: *(int*)d->dst = 1; break;
	case ARG_I32: *(int32_t*)d->dst = (int32_t)strtol(text, 0, 10); break;
	case ARG_SIZE: *(int32_t*)d->dst = (int32_t)(with_suffix(text) + .499); break;
	case ARG_I64SIZE: *(int64_t*)d->dst = (int64_t)(with_suffix(text) + .499); break;
	case ARG_F64: *(double*)d->dst = strtod(text, 0); break;
	case ARG_TEXT: *(const char**)d->dst = text; break;
	}
}

/* returns the index of the first positional argument, or -1 after reporting a bad option */
static int arg_scan(int argc, char **argv, const struct arg_def *defs, int n_defs)
{
	int i = 1;
	for (; i < argc; ++i) {
		const char *a = argv[i];
		if (a[0] != '-' || a[1] == 0) break;
		if (a[1] == '-' && a[2] == 0) { ++i; break; }
		for (const char *q = a + 1; *q; ++q) {               /* flags may be grouped; a valued option ends the group */
			const struct arg_def *d = 0;
			for (int j = 0; j < n_defs; ++j) if (defs[j].letter == *q) d = &defs[j];
			if (!d) { fprintf(stderr, "yak-amd: unknown option -%c\n", *q); return -1; }
			if (d->kind == ARG_FLAG) { arg_store(d, 0); continue; }
			if (q[1]) arg_store(d, q + 1);
			else if (i + 1 < argc) arg_store(d, argv[++i]);
			else { fprintf(stderr, "yak-amd: option -%c needs a value\n", *q); return -1; }
			break;
		}
	}
	return i;
}

static void arg_help(const char *synopsis, const struct arg_def *defs, int n_defs)
{
	fprintf(stderr, "usage: yak-amd %s\n", synopsis);
	for (int j = 0; j < n_defs; ++j) fprintf(stderr, "    -%c%s  %s\n", defs[j].letter, defs[j].kind == ARG_FLAG ? "     " : " VAL ", defs[j].what);
}

/* ---- count ---- */
static int cmd_count(int argc, char **argv)
{
	yak_copt_t o;
	const char *out = 0;
	yak_copt_init(&o);
	const struct arg_def defs[] = {
		{ 'k', ARG_I32, &o.k, "k-mer length, below 64 (counts are approximate from 32 on)" },
		{ 'p', ARG_I32, &o.pre, "bits of the hash that pick the sub-table" },
		{ 'b', ARG_I32, &o.bf_shift, "log2 bits of the bloom prefilter; 0 = one pass, singletons kept" },
		{ 'H', ARG_SIZE, &o.bf_n_hash, "probes per bloom lookup" },
		{ 't', ARG_I32, &o.n_thread, "host threads (parser)" },
		{ 'K', ARG_I64SIZE, &o.chunk_size, "bases per host batch" },
		{ 'o', ARG_TEXT, &out, "write the table (.yak) here" },
	};
	const int nd = (int)(sizeof(defs) / sizeof(defs[0]));
	const int first = arg_scan(argc, argv, defs, nd);
	if (first < 0 || first >= argc) { arg_help("count [options] <reads.fa|fq[.gz]> [second-pass reads]", defs, nd); return 1; }
	if (o.pre < YAK_COUNTER_BITS || o.k < 1 || o.k >= 64) { fprintf(stderr, "yak-amd count: need 1 <= k < 64 and p >= %d\n", YAK_COUNTER_BITS); return 1; }
	if (o.k >= 32) fprintf(stderr, "yak-amd count: k >= 32 uses the 64-bit sum hash: counts are approximate\n");
	const char *pass1 = argv[first], *pass2 = first + 1 < argc ? argv[first + 1] : argv[first];
	yak_ch_t *tab = yak_count(pass1, &o, 0);
	if (!tab) { fprintf(stderr, "yak-amd count: no table (unreadable input, or no MI355X)\n"); return 2; }
	if (o.bf_shift > 0) {                                    /* filtered mode: pass 1 picked the keys, pass 2 counts them */
		yak_ch_destroy_bf(tab);
		yak_ch_clear(tab, o.n_thread);
		if (!yak_count(pass2, &o, tab)) { yak_ch_destroy(tab); return 2; }
		yak_ch_shrink(tab, 2, YAK_MAX_COUNT, o.n_thread);
		fprintf(stderr, "[M::yak-amd] %ld distinct k-mers after shrinking\n", (long)tab->tot);
	}
	int rc = 0;
	if (out && yak_ch_dump(tab, out) != 0) { fprintf(stderr, "yak-amd count: cannot write %s\n", out); rc = 3; }
	yak_ch_destroy(tab);
	return rc;
}

/* ---- qv ---- */
static const char *const qv_legend[] = {                     /* output format of `yak qv` (reference main.c:196-201) */
	"CC\tCT  kmer_occurrence    short_read_kmer_count  raw_input_kmer_count  adjusted_input_kmer_count",
	"CC\tFR  fpr_lower_bound    fpr_upper_bound",
	"CC\tER  total_input_kmers  adjusted_error_kmers",
	"CC\tCV  coverage",
	"CC\tQV  raw_quality_value  adjusted_quality_value",
	"CC",
};

static int cmd_qv(int argc, char **argv)
{
	yak_qopt_t o;
	yak_qopt_init(&o);
	const struct arg_def defs[] = {
		{ 'l', ARG_SIZE, &o.min_len, "skip sequences shorter than this" },
		{ 'f', ARG_F64, &o.min_frac, "skip sequences with a smaller share of known k-mers" },
		{ 'e', ARG_F64, &o.fpr, "assumed false-positive rate of \"absent\"" },
		{ 'p', ARG_FLAG, &o.print_each, "one SQ line per sequence" },
		{ 'E', ARG_FLAG, &o.print_err_kmer, "one EK line per absent k-mer run" },
		{ 't', ARG_I32, &o.n_threads, "host threads" },
		{ 'K', ARG_I64SIZE, &o.chunk_size, "bases per device batch" },
	};
	const int nd = (int)(sizeof(defs) / sizeof(defs[0]));
	const int first = arg_scan(argc, argv, defs, nd);
	if (first < 0 || first + 1 >= argc) { arg_help("qv [options] <table.yak> <sequences.fa>", defs, nd); return 1; }
	yak_ch_t *tab = yak_ch_restore(argv[first]);
	if (!tab) { fprintf(stderr, "yak-amd qv: cannot load %s (or no MI355X)\n", argv[first]); return 2; }
	static int64_t in_table[YAK_N_COUNTS], in_seqs[YAK_N_COUNTS];
	static yak_qstat_t st;
	const int k = tab->k;
	yak_ch_hist(tab, in_table, o.n_threads);
	for (size_t i = 0; i < sizeof(qv_legend) / sizeof(qv_legend[0]); ++i) puts(qv_legend[i]);
	yak_qv(&o, argv[first + 1], tab, in_seqs);
	yak_qv_solve(in_table, in_seqs, k, o.fpr, &st);
	for (int c = YAK_N_COUNTS; c-- > 0;) printf("CT\t%d\t%ld\t%ld\t%.3f\n", c, (long)in_table[c], (long)in_seqs[c], st.adj_cnt[c]);
	printf("FR\t%.3g\t%.3g\n", st.fpr_lower, st.fpr_upper);
	printf("ER\t%ld\t%.3f\n", (long)st.tot, st.err);
	printf("CV\t%.3f\n", st.cov);
	printf("QV\t%.3f\t%.3f\n", st.qv_raw, st.qv);
	yak_ch_destroy(tab);
	return 0;
}

int main(int argc, char **argv)
{
	static const struct { const char *name; int (*run)(int, char**); const char *what; } cmds[] = {
		{ "count", cmd_count, "count k-mers on the GPU, write a .yak table" },
		{ "qv", cmd_qv, "look the k-mers of sequences up in a .yak table" },
	};
	/* -X name=value (anywhere on the line, any number of times): a test switch of the library (yakamd_test_set) -- tests force code paths with it */
	for (int i = 1; i + 1 < argc; ) {
		char *eq = strcmp(argv[i], "-X") == 0 ? strchr(argv[i + 1], '=') : 0;
		if (!eq) { ++i; continue; }
		*eq = 0;
		yakamd_test_set(argv[i + 1], atoll(eq + 1));
		memmove(argv + i, argv + i + 2, (size_t)(argc - i - 2) * sizeof(char*));
		argc -= 2;
	}
	if (argc >= 2)
		for (size_t i = 0; i < sizeof(cmds) / sizeof(cmds[0]); ++i)
			if (strcmp(argv[1], cmds[i].name) == 0) return cmds[i].run(argc - 1, argv + 1);
	fprintf(stderr, "yak-amd: driver of libyak_amd.so (lh3/yak's C API on MI355X)\n");
	for (size_t i = 0; i < sizeof(cmds) / sizeof(cmds[0]); ++i) fprintf(stderr, "    yak-amd %-6s %s\n", cmds[i].name, cmds[i].what);
	return 1;
}
