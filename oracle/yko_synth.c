/*
 * yko_synth.c -- ORACLE (test infrastructure only): `yak count` without a filter over a SYNTHETIC read stream that is too
 * large to write down (BASELINE configs[2]: 600 M x 150 bp = 90 GB of bases), restricted to a few prefix ranges.
 *
 * The stream is tools/yaksynth.c's (reads first, first + 1, ... in order, each followed by '\n'); it is generated chunk by
 * chunk, never stored.  Per chunk the reads are dealt to the threads in contiguous slices; every thread extracts its
 * slice's k-mers with the oracle's own yko_extract (count.c:28-60) into per-prefix lists of its own, and the lists of a
 * prefix are then inserted in thread order = stream order by yko_ch_insert_list (htab.c:51-78), one prefix per worker at a
 * time -- the reference's own parallel scheme (count.c:129-143: kt_for over prefixes; a sub-table is a function of its own
 * put-calls in stream order, whatever -t and -K are).  Only the prefixes of the ranges asked for are inserted (every
 * sub-table is independent of the others), so a range comes out exactly as in a full run: yko_ch_dump_range writes its
 * bytes, which are what rank r of an N-GPU prefix-sharded job owns.
 *
 *   yko_synth -n reads -l len -g genome -s seed -e err -N nrate -k k -t threads -c chunk_reads -R lo:hi [-R lo:hi ...] -o out_prefix
 *   -> out_prefix.<lo>-<hi>.part per range ({capacity, size, keys in slot order} per sub-table; no .yak header) and one
 *      line "RANGE lo hi distinct instances" per range on stdout.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <pthread.h>
#include "yko.h"

int64_t yaksynth_reads(uint8_t *out, int64_t n_reads, int read_len, int64_t genome_len, uint64_t seed,
                       double err, double nrate, int64_t first_read, int n_threads);

#define MAX_R 16
#define MAX_T 64

typedef struct {
	int tid, nt, k, pre, rlen;
	int64_t genome, first, n_reads;          /* this chunk */
	uint64_t seed;
	double err, nrate;
	uint8_t *img;                             /* the thread's slice of the chunk image */
	yko_kbuf_t *buf;                          /* P lists */
} gen_t;

static yko_ch_t *g_h;
static gen_t g_gen[MAX_T];
static int g_nt, g_P;
static unsigned char g_want[1 << 16];
static volatile int g_next;
static int64_t g_ins[MAX_T];

static void *gen_worker(void *a)
{
	gen_t *g = (gen_t*)a;
	const int64_t r0 = g->n_reads * g->tid / g->nt, r1 = g->n_reads * (g->tid + 1) / g->nt;
	int64_t r;
	if (r1 <= r0) return 0;
	yaksynth_reads(g->img, r1 - r0, g->rlen, g->genome, g->seed, g->err, g->nrate, g->first + r0, 1);
	for (r = 0; r < r1 - r0; ++r)                      /* a read = one record of count.c:93-96; reads shorter than k are dropped there, none is here */
		if (g->rlen >= g->k) yko_extract(g->buf, g->k, g->pre, g->rlen, (const char*)g->img + r * (g->rlen + 1));
	return 0;
}

static void *ins_worker(void *a)
{
	const int me = (int)(intptr_t)a;
	for (;;) {
		const int p = __sync_fetch_and_add(&g_next, 1);
		int t;
		if (p >= g_P) break;
		for (t = 0; t < g_nt; ++t) {
			yko_kbuf_t *b = &g_gen[t].buf[p];
			if (g_want[p] && b->n) {
				int64_t off = 0;
				while (off < b->n) {                   /* yak_ch_insert_list takes an int count */
					int64_t w = b->n - off > (1 << 30) ? (1 << 30) : b->n - off;
					g_ins[me] += yko_ch_insert_list(g_h, 1, (int)w, b->a + off);
					off += w;
				}
			}
			b->n = 0;
		}
	}
	return 0;
}

int main(int argc, char *argv[])
{
	int64_t n = 1000000, g = 5000000, chunk = 2000000, done;
	int l = 150, k = 31, pre = 10, nt = 8, c, nr = 0, rlo[MAX_R], rhi[MAX_R], i, t;
	uint64_t seed = 42;
	double e = 0.001, nrate = 0.0005;
	const char *out = 0;
	pthread_t th[MAX_T];
	while ((c = getopt(argc, argv, "n:l:g:s:e:N:k:p:t:c:R:o:")) >= 0) {
		if (c == 'n') n = atoll(optarg); else if (c == 'l') l = atoi(optarg); else if (c == 'g') g = atoll(optarg);
		else if (c == 's') seed = strtoull(optarg, 0, 10); else if (c == 'e') e = atof(optarg); else if (c == 'N') nrate = atof(optarg);
		else if (c == 'k') k = atoi(optarg); else if (c == 'p') pre = atoi(optarg); else if (c == 't') nt = atoi(optarg);
		else if (c == 'c') chunk = atoll(optarg); else if (c == 'o') out = optarg;
		else if (c == 'R') { if (nr == MAX_R || sscanf(optarg, "%d:%d", &rlo[nr], &rhi[nr]) != 2) return 1; ++nr; }
	}
	if (nr == 0 || out == 0 || nt < 1 || nt > MAX_T || pre < YKO_COUNTER_BITS || pre > 16 || k >= 64) {
		fprintf(stderr, "Usage: yko_synth -n reads -l len -g genome -s seed -e err -N nrate -k k -t threads -c chunk_reads -R lo:hi [-R ...] -o out_prefix\n");
		return 1;
	}
	g_P = 1 << pre; g_nt = nt;
	for (i = 0; i < nr; ++i) { int p; if (rlo[i] < 0 || rlo[i] >= rhi[i] || rhi[i] > g_P) return 1; for (p = rlo[i]; p < rhi[i]; ++p) g_want[p] = 1; }
	yko_set_prefix_mask(g_want);                                 /* k-mers of other prefixes are not even listed */
	g_h = yko_ch_init(k, pre, 0, 0);
	for (t = 0; t < nt; ++t) {
		g_gen[t].tid = t; g_gen[t].nt = nt; g_gen[t].k = k; g_gen[t].pre = pre; g_gen[t].rlen = l; g_gen[t].genome = g; g_gen[t].seed = seed;
		g_gen[t].err = e; g_gen[t].nrate = nrate;
		g_gen[t].img = (uint8_t*)malloc((size_t)(chunk / nt + 2) * (l + 1));
		g_gen[t].buf = (yko_kbuf_t*)calloc(g_P, sizeof(yko_kbuf_t));
	}
	for (done = 0; done < n; done += chunk) {
		const int64_t m = n - done < chunk ? n - done : chunk;
		for (t = 0; t < nt; ++t) { g_gen[t].first = done; g_gen[t].n_reads = m; pthread_create(&th[t], 0, gen_worker, &g_gen[t]); }
		for (t = 0; t < nt; ++t) pthread_join(th[t], 0);
		g_next = 0;
		for (t = 0; t < nt; ++t) pthread_create(&th[t], 0, ins_worker, (void*)(intptr_t)t);
		for (t = 0; t < nt; ++t) pthread_join(th[t], 0);
		if ((done / chunk) % 10 == 0) fprintf(stderr, "[yko_synth] %ld of %ld reads\n", (long)(done + m), (long)n);
	}
	for (t = 0; t < nt; ++t) g_h->tot += (uint64_t)g_ins[t];
	for (i = 0; i < nr; ++i) {
		char fn[4096];
		int64_t distinct = 0, inst = 0;
		int p;
		snprintf(fn, sizeof(fn), "%s.%d-%d.part", out, rlo[i], rhi[i]);
		/* the 16-byte .yak header goes with range 0 in yko_ch_dump_range: a rank's share is the sub-tables alone, so dump [lo, hi) as two calls when lo == 0 */
		if (rlo[i] == 0) {
			char f2[4200];
			FILE *fa, *fb; int ch_;
			snprintf(f2, sizeof(f2), "%s.hdr", fn);
			if (yko_ch_dump_range(g_h, f2, 0, rhi[i]) != 0) return 2;
			fa = fopen(f2, "rb"); fb = fopen(fn, "wb");
			if (!fa || !fb) return 2;
			fseek(fa, 16, SEEK_SET);
			{ static char blk[1 << 20]; size_t got; while ((got = fread(blk, 1, sizeof(blk), fa)) > 0) fwrite(blk, 1, got, fb); }
			(void)ch_;
			fclose(fa); fclose(fb); remove(f2);
		} else if (yko_ch_dump_range(g_h, fn, rlo[i], rhi[i]) != 0) return 2;
		for (p = rlo[i]; p < rhi[i]; ++p) {
			uint32_t cap, size, s;
			const yko_set_t *st = g_h->h[p].h;
			yko_ch_subtable(g_h, p, &cap, &size);
			distinct += size;
			for (s = 0; s < cap; ++s) if (st->used[s >> 5] >> (s & 31) & 1) inst += (int64_t)(st->keys[s] & YKO_MAX_COUNT);
		}
		printf("RANGE %d %d %ld %ld\n", rlo[i], rhi[i], (long)distinct, (long)inst);
	}
	return 0;
}
