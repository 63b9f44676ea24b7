/*
 * yko_main.c -- ORACLE command line (test infrastructure only): `yko count` takes the same
 * options as `yak count` (main.c:13-64) and writes the same .yak file, using the CPU restatement;
 * `yko qv` is the counting step of `yak qv`.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "yko.h"

static int64_t parse_num(const char *s)                      /* yak-priv.h:75-84 */
{
	char *p;
	double x = strtod(s, &p);
	if (*p == 'G' || *p == 'g') x *= 1e9;
	else if (*p == 'M' || *p == 'm') x *= 1e6;
	else if (*p == 'K' || *p == 'k') x *= 1e3;
	return (int64_t)(x + .499);
}

/* `yko qv`: the counting step of `yak qv` (main.c:163-215 without the statistics of yak_qv_solve):
 * EK / SQ lines as the reference prints them, then "CT\t<count>\t<table k-mers>\t<input k-mers>" */
static int main_qv(int argc, char *argv[])
{
	yko_qopt_t opt;
	yko_ch_t *ch;
	int64_t cnt[1 << YKO_COUNTER_BITS], hist[1 << YKO_COUNTER_BITS];
	int c, i;
	yko_qopt_init(&opt);
	while ((c = getopt(argc, argv, "K:t:l:f:pe:E")) >= 0) {
		if (c == 'K') opt.chunk_size = parse_num(optarg);
		else if (c == 'l') opt.min_len = (int32_t)parse_num(optarg);
		else if (c == 'f') opt.min_frac = atof(optarg);
		else if (c == 't') opt.n_threads = atoi(optarg);
		else if (c == 'p') opt.print_each = 1;
		else if (c == 'E') opt.print_err_kmer = 1;
		else if (c == 'e') opt.fpr = atof(optarg);
	}
	if (argc - optind < 2) return 1;
	ch = yko_ch_restore(argv[optind]);
	if (!ch) return 2;
	yko_ch_hist(ch, hist);
	if (yko_qv(&opt, argv[optind + 1], ch, cnt, stdout) != 0) return 2;
	for (i = (1 << YKO_COUNTER_BITS) - 1; i >= 0; --i) printf("CT\t%d\t%ld\t%ld\n", i, (long)hist[i], (long)cnt[i]);
	yko_ch_destroy(ch);
	return 0;
}

int main(int argc, char *argv[])
{
	yko_copt_t opt;
	yko_ch_t *h;
	const char *out = 0;
	int c, rlo = -1, rhi = -1;
	if (argc >= 2 && strcmp(argv[1], "qv") == 0) return main_qv(argc - 1, argv + 1);
	if (argc < 2 || strcmp(argv[1], "count") != 0) {
		fprintf(stderr, "Usage: yko count [-k31] [-p10] [-K chunk] [-t thr] [-b bloom_bits] [-H n_hash] [-R lo:hi (sub-tables of this prefix range only; the outputs of consecutive ranges concatenate to the .yak file)] [-o out.yak] <in.fa> [in2.fa]\n");
		return 1;
	}
	yko_copt_init(&opt);
	--argc; ++argv;
	while ((c = getopt(argc, argv, "k:p:K:t:b:H:o:R:")) >= 0) {
		if (c == 'k') opt.k = atoi(optarg);
		else if (c == 'p') opt.pre = atoi(optarg);
		else if (c == 'K') opt.chunk_size = parse_num(optarg);
		else if (c == 't') opt.n_thread = atoi(optarg);
		else if (c == 'b') opt.bf_shift = atoi(optarg);
		else if (c == 'H') opt.bf_n_hash = (int)parse_num(optarg);
		else if (c == 'o') out = optarg;
		else if (c == 'R') { if (sscanf(optarg, "%d:%d", &rlo, &rhi) != 2) return 1; }
	}
	if (argc - optind < 1 || opt.pre < YKO_COUNTER_BITS || opt.k >= 64) return 1;   /* main.c:30-52 */
	if (rlo >= 0 && (rlo >= rhi || rhi > (1 << opt.pre))) return 1;
	if (rlo >= 0) yko_set_prefix_range(rlo, rhi);
	h = yko_count_protocol_file(argv[optind], argc - optind >= 2 ? argv[optind + 1] : 0, &opt);
	if (h == 0) return 2;
	fprintf(stderr, "[yko] %ld distinct k-mers\n", (long)h->tot);
	if (out && rlo >= 0) yko_ch_dump_range(h, out, rlo, rhi);
	else if (out) yko_ch_dump(h, out);
	yko_ch_destroy(h);
	return 0;
}
