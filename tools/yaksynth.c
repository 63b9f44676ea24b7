/*
 * yaksynth.c -- deterministic synthetic read generator shared by tests, bench.py and the CLIs.
 *
 * Model: a random genome of G bases, i.i.d. uniform over ACGT; N reads of L bp,
 * start uniform in [0, G-L], reverse-complemented with p = 0.5, each base substituted with
 * probability `err` (by one of the three other bases) and replaced by 'N' with probability
 * `nrate`.  Everything is counter based (splitmix64 of (seed, index)), so any slice of reads can
 * be produced independently and identically on any machine / thread count.
 *
 * Output "memory image": each read is L ASCII bases followed by one '\n'.  The engine and the oracle
 * treat every non-ACGT byte as a k-mer break, so this image is equivalent to a FASTA/FASTQ file of
 * the same reads in the same order.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <unistd.h>

static inline uint64_t mix64(uint64_t z)
{
	z += 0x9e3779b97f4a7c15ULL;
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	return z ^ (z >> 31);
}
static inline uint64_t rnd(uint64_t seed, uint64_t stream, uint64_t idx)
{
	return mix64(mix64(seed * 0x100000001b3ULL + stream) ^ idx);
}

static inline int genome_base(uint64_t seed, int64_t i)
{
	uint64_t w = rnd(seed, 1, (uint64_t)i >> 5);
	return (int)(w >> (2 * (i & 31)) & 3);
}

void yaksynth_genome(uint8_t *g, int64_t glen, uint64_t seed)
{
	int64_t i;
	for (i = 0; i < glen; ++i) g[i] = "ACGT"[genome_base(seed, i)];
}

typedef struct {
	uint8_t *out;
	int64_t r0, r1, first_read, glen;
	int rlen;
	uint64_t seed;
	uint32_t err24, n24;
} job_t;

static void gen_range(const job_t *j)
{
	int64_t r;
	for (r = j->r0; r < j->r1; ++r) {
		uint64_t rid = (uint64_t)(j->first_read + r);
		uint64_t h = rnd(j->seed, 2, rid);
		int64_t start = (int64_t)((h >> 1) % (uint64_t)(j->glen - j->rlen + 1));
		int rev = (int)(h & 1), p;
		uint8_t *o = j->out + r * (int64_t)(j->rlen + 1);
		for (p = 0; p < j->rlen; ++p) {
			uint64_t u = rnd(j->seed, 3, rid * (uint64_t)j->rlen + (uint64_t)p);
			int b = rev ? 3 - genome_base(j->seed, start + j->rlen - 1 - p) : genome_base(j->seed, start + p);
			if ((uint32_t)(u & 0xffffff) < j->err24) b = (b + 1 + (int)((u >> 24 & 0xff) % 3)) & 3;
			o[p] = ((uint32_t)(u >> 32 & 0xffffff) < j->n24) ? 'N' : (uint8_t)"ACGT"[b];
		}
		o[j->rlen] = '\n';
	}
}

static void *gen_thread(void *a) { gen_range((const job_t*)a); return 0; }

/* writes n_reads * (read_len + 1) bytes; returns that size.  `first_read` offsets the read index
 * (rank r of an N-way shard passes r * reads_per_rank) */
int64_t yaksynth_reads(uint8_t *out, int64_t n_reads, int read_len, int64_t genome_len, uint64_t seed,
                       double err, double nrate, int64_t first_read, int n_threads)
{
	job_t jobs[256];
	pthread_t tid[256];
	int t;
	if (n_threads < 1) n_threads = 1;
	if (n_threads > 256) n_threads = 256;
	if (genome_len < read_len) return -1;
	for (t = 0; t < n_threads; ++t) {
		jobs[t].out = out; jobs[t].first_read = first_read; jobs[t].glen = genome_len;
		jobs[t].rlen = read_len; jobs[t].seed = seed;
		jobs[t].err24 = (uint32_t)(err * 16777216.0 + 0.5); jobs[t].n24 = (uint32_t)(nrate * 16777216.0 + 0.5);
		jobs[t].r0 = n_reads * t / n_threads; jobs[t].r1 = n_reads * (t + 1) / n_threads;
	}
	for (t = 1; t < n_threads; ++t) pthread_create(&tid[t], 0, gen_thread, &jobs[t]);
	gen_range(&jobs[0]);
	for (t = 1; t < n_threads; ++t) pthread_join(tid[t], 0);
	return n_reads * (int64_t)(read_len + 1);
}

/* "assembly" image (BASELINE configs[3]): contig i is genome[i * contig_len, (i + 1) * contig_len), every
 * contig followed by one '\n' -- a gap-free tiling of an n_contigs * contig_len genome, so nearly every
 * k-mer is distinct (the long-contig, singletons-kept regime).  Bytes [out, out + n * (len + 1)). */
typedef struct { uint8_t *out; int64_t b0, b1, clen; uint64_t seed; } tjob_t;
static void *tile_thread(void *a)
{
	const tjob_t *j = (const tjob_t*)a;
	int64_t x;
	for (x = j->b0; x < j->b1; ++x) {                    /* x = byte index in the image */
		const int64_t c = x / (j->clen + 1), o = x % (j->clen + 1);
		j->out[x] = o == j->clen ? '\n' : (uint8_t)"ACGT"[genome_base(j->seed, c * j->clen + o)];
	}
	return 0;
}
int64_t yaksynth_tiles(uint8_t *out, int64_t n_contigs, int64_t contig_len, uint64_t seed, int n_threads)
{
	tjob_t jobs[256];
	pthread_t tid[256];
	const int64_t tot = n_contigs * (contig_len + 1);
	int t;
	if (n_threads < 1) n_threads = 1;
	if (n_threads > 256) n_threads = 256;
	for (t = 0; t < n_threads; ++t) { jobs[t].out = out; jobs[t].clen = contig_len; jobs[t].seed = seed; jobs[t].b0 = tot * t / n_threads; jobs[t].b1 = tot * (t + 1) / n_threads; }
	for (t = 1; t < n_threads; ++t) pthread_create(&tid[t], 0, tile_thread, &jobs[t]);
	tile_thread(&jobs[0]);
	for (t = 1; t < n_threads; ++t) pthread_join(tid[t], 0);
	return tot;
}

#ifdef YAKSYNTH_MAIN
int main(int argc, char *argv[])
{
	int64_t n = 1000, g = 100000, i;
	int l = 150, c, fasta = 0, thr = 4, tiles = 0, wrap = 0;
	int64_t ll = 150;
	uint64_t seed = 42;
	double e = 0.005, nr = 0.0005;
	uint8_t *buf;
	FILE *fp = stdout;
	while ((c = getopt(argc, argv, "n:l:g:s:e:N:o:at:Tw:")) >= 0) {
		if (c == 'n') n = atoll(optarg); else if (c == 'l') { ll = atoll(optarg); l = (int)ll; }
		else if (c == 'T') tiles = fasta = 1; else if (c == 'w') wrap = atoi(optarg);
		else if (c == 'g') g = atoll(optarg); else if (c == 's') seed = strtoull(optarg, 0, 10);
		else if (c == 'e') e = atof(optarg); else if (c == 'N') nr = atof(optarg);
		else if (c == 'a') fasta = 1; else if (c == 't') thr = atoi(optarg);
		else if (c == 'o') { fp = fopen(optarg, "wb"); if (!fp) { perror(optarg); return 1; } }
	}
	if (tiles) {                                        /* -T: n contigs of l bases tiling a genome of n * l bases, FASTA, -w columns per line (0 = one line) */
		buf = (uint8_t*)malloc((size_t)n * (ll + 1));
		yaksynth_tiles(buf, n, ll, seed, thr);
		for (i = 0; i < n; ++i) {
			const uint8_t *sq = buf + i * (ll + 1);
			int64_t o;
			fprintf(fp, ">ctg%ld\n", (long)i);
			if (wrap <= 0) fwrite(sq, 1, ll + 1, fp);
			else for (o = 0; o < ll; o += wrap) { fwrite(sq + o, 1, ll - o < wrap ? ll - o : wrap, fp); fputc('\n', fp); }
		}
		if (fp != stdout) fclose(fp);
		free(buf);
		return 0;
	}
	buf = (uint8_t*)malloc((size_t)n * (l + 1));
	if (yaksynth_reads(buf, n, l, g, seed, e, nr, 0, thr) < 0) return 1;
	{
		char *qual = (char*)malloc(l + 2);
		memset(qual, 'I', l); qual[l] = '\n';
		for (i = 0; i < n; ++i) {
			fprintf(fp, "%cr%ld\n", fasta ? '>' : '@', (long)i);
			fwrite(buf + i * (int64_t)(l + 1), 1, l + 1, fp);
			if (!fasta) { fputs("+\n", fp); fwrite(qual, 1, l + 1, fp); }
		}
		free(qual);
	}
	if (fp != stdout) fclose(fp);
	free(buf);
	return 0;
}
#endif
