/*
 * main.c -- `yak-amd count` (and `yak-amd qv`): the caller side of the hot path, i.e. what reference main.c:13-64
 * (main_count) does, written in C against include/yak.h only.  It exists to show that the
 * library is a drop-in: the protocol below is the reference's, line for line in meaning
 * (count -> [destroy_bf, clear, second pass, shrink] -> dump).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "yak.h"

static long long parse_num(const char *s)                    /* K/M/G suffixes as yak-priv.h:75-84 */
{
	char *p;
	double x = strtod(s, &p);
	if (*p == 'G' || *p == 'g') x *= 1e9;
	else if (*p == 'M' || *p == 'm') x *= 1e6;
	else if (*p == 'K' || *p == 'k') x *= 1e3;
	return (long long)(x + .499);
}

/* `yak-amd qv`: reference main.c:163-215 (main_qv) -- restore, histogram, the device lookup of every
 * k-mer of <seq.fa>, then the host statistics of yak_qv_solve; same output lines as the reference. */
static int main_qv(int argc, char *argv[])
{
	yak_qopt_t opt;
	yak_ch_t *ch;
	int64_t cnt[YAK_N_COUNTS], hist[YAK_N_COUNTS];
	static yak_qstat_t qs;
	int c, i, kmer;
	yak_qopt_init(&opt);
	while ((c = getopt(argc, argv, "K:t:l:f:pe:E")) >= 0) {
		if (c == 'K') opt.chunk_size = parse_num(optarg);
		else if (c == 'l') opt.min_len = (int32_t)parse_num(optarg);
		else if (c == 'f') opt.min_frac = atof(optarg);
		else if (c == 't') opt.n_threads = atoi(optarg);
		else if (c == 'p') opt.print_each = 1;
		else if (c == 'E') opt.print_err_kmer = 1;
		else if (c == 'e') opt.fpr = atof(optarg);
	}
	if (argc - optind < 2) {
		fprintf(stderr, "Usage: yak-amd qv [-l min_len] [-f min_frac] [-p] [-E] [-K batch] <kmer.hash> <seq.fa>\n");
		return 1;
	}
	ch = yak_ch_restore(argv[optind]);
	if (ch == 0) { fprintf(stderr, "ERROR: failed to load '%s' (or no MI355X visible)\n", argv[optind]); return 1; }
	kmer = ch->k;
	yak_ch_hist(ch, hist, opt.n_threads);
	printf("CC\tCT  kmer_occurrence    short_read_kmer_count  raw_input_kmer_count  adjusted_input_kmer_count\n");
	printf("CC\tFR  fpr_lower_bound    fpr_upper_bound\n");
	printf("CC\tER  total_input_kmers  adjusted_error_kmers\n");
	printf("CC\tCV  coverage\n");
	printf("CC\tQV  raw_quality_value  adjusted_quality_value\n");
	printf("CC\n");
	yak_qv(&opt, argv[optind + 1], ch, cnt);
	yak_qv_solve(hist, cnt, kmer, opt.fpr, &qs);
	for (i = YAK_N_COUNTS - 1; i >= 0; --i)
		printf("CT\t%d\t%ld\t%ld\t%.3f\n", i, (long)hist[i], (long)cnt[i], qs.adj_cnt[i]);
	printf("FR\t%.3g\t%.3g\n", qs.fpr_lower, qs.fpr_upper);
	printf("ER\t%ld\t%.3f\n", (long)qs.tot, qs.err);
	printf("CV\t%.3f\n", qs.cov);
	printf("QV\t%.3f\t%.3f\n", qs.qv_raw, qs.qv);
	yak_ch_destroy(ch);
	return 0;
}

int main(int argc, char *argv[])
{
	yak_copt_t opt;
	yak_ch_t *h;
	const char *fn_out = 0;
	int c;
	if (argc >= 2 && strcmp(argv[1], "qv") == 0) return main_qv(argc - 1, argv + 1);
	if (argc < 2 || strcmp(argv[1], "count") != 0) {
		fprintf(stderr, "Usage: yak-amd count [options] <in.fa> [in.fa]\n       yak-amd qv [options] <kmer.hash> <seq.fa>\n");
		return 1;
	}
	--argc, ++argv;
	yak_copt_init(&opt);
	while ((c = getopt(argc, argv, "k:p:K:t:b:H:o:")) >= 0) {
		if (c == 'k') opt.k = atoi(optarg);
		else if (c == 'p') opt.pre = atoi(optarg);
		else if (c == 'K') opt.chunk_size = parse_num(optarg);
		else if (c == 't') opt.n_thread = atoi(optarg);
		else if (c == 'b') opt.bf_shift = atoi(optarg);
		else if (c == 'H') opt.bf_n_hash = (int)parse_num(optarg);
		else if (c == 'o') fn_out = optarg;
	}
	if (argc - optind < 1) {
		fprintf(stderr, "Usage: yak-amd count [options] <in.fa> [in.fa]\n");
		fprintf(stderr, "Options:\n");
		fprintf(stderr, "  -k INT     k-mer size [%d]\n", opt.k);
		fprintf(stderr, "  -p INT     prefix length [%d]\n", opt.pre);
		fprintf(stderr, "  -b INT     set Bloom filter size to 2**INT bits; 0 to disable [%d]\n", opt.bf_shift);
		fprintf(stderr, "  -H INT     use INT hash functions for Bloom filter [%d]\n", opt.bf_n_hash);
		fprintf(stderr, "  -t INT     number of host worker threads [%d]\n", opt.n_thread);
		fprintf(stderr, "  -o FILE    dump the count hash table to FILE []\n");
		fprintf(stderr, "  -K INT     chunk size [100m]\n");
		return 1;
	}
	if (opt.pre < YAK_COUNTER_BITS) { fprintf(stderr, "ERROR: -p should be at least %d\n", YAK_COUNTER_BITS); return 1; }
	if (opt.k >= 64) { fprintf(stderr, "ERROR: -k must be smaller than 64\n"); return 1; }
	else if (opt.k >= 32) fprintf(stderr, "WARNING: counts are inexact if -k is greater than 31\n");
	h = yak_count(argv[optind], &opt, 0);
	if (h == 0) { fprintf(stderr, "ERROR: counting failed (input unreadable or no MI355X available)\n"); return 2; }
	if (opt.bf_shift > 0) {
		yak_ch_destroy_bf(h);
		yak_ch_clear(h, opt.n_thread);
		h = yak_count(argc - optind >= 2 ? argv[optind + 1] : argv[optind], &opt, h);
		if (h == 0) return 2;
		yak_ch_shrink(h, 2, YAK_MAX_COUNT, opt.n_thread);
		fprintf(stderr, "[M::%s] %ld distinct k-mers after shrinking\n", __func__, (long)h->tot);
	}
	if (fn_out) yak_ch_dump(h, fn_out);
	yak_ch_destroy(h);
	return 0;
}
