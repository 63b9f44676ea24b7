/*
 * yko_count.c -- ORACLE (test infrastructure only).  CPU restatement of yak's counting driver:
 * FASTA/FASTQ record reader (behaviour of kseq.h:192-232 as used by count.c:88-110), canonical
 * k-mer extraction (count.c:28-60), per-prefix bucketing (count.c:17-26) and the serial
 * equivalent of the 3-step pipeline (count.c:85-166).  The pipeline's only observable contract
 * is ordering: within each sub-table the put-calls happen in input-stream order, whatever -t and
 * -K are (kthread.c:107-112); a serial loop chunk by chunk reproduces exactly that.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <zlib.h>
#include "yko.h"

/* ------------------------------------------------------------------ k-mer extraction */
static const unsigned char *g_pmask;                         /* yko_set_prefix_mask: k-mers of unwanted sub-tables are not listed at all */
void yko_set_prefix_mask(const unsigned char *want) { g_pmask = want; }

static inline void kbuf_push(yko_kbuf_t *b, uint64_t y)      /* count.c:17-26 */
{
	if (b->n == b->m) {
		b->m = b->m < 8 ? 8 : b->m + (b->m >> 1);
		b->a = (uint64_t*)realloc(b->a, (size_t)b->m * 8);
	}
	b->a[b->n++] = y;
}

static void extract_short(yko_kbuf_t *buf, int k, int pre, int64_t len, const char *seq) /* count.c:28-43 */
{
	const uint64_t mask = (1ULL << 2 * k) - 1, pm = (1ULL << pre) - 1;
	const int shift = 2 * (k - 1);
	uint64_t fw = 0, rv = 0;
	int64_t i;
	int run = 0;                                             /* #consecutive ACGT bases seen */
	for (i = 0; i < len; ++i) {
		int c = yko_nt4[(uint8_t)seq[i]];
		if (c >= 4) { run = 0; fw = rv = 0; continue; }      /* ambiguous base: restart */
		fw = (fw << 2 | (uint64_t)c) & mask;
		rv = rv >> 2 | (uint64_t)(3 - c) << shift;
		if (++run >= k) {
			uint64_t h = yko_hash64(fw < rv ? fw : rv, mask);
			if (g_pmask == 0 || g_pmask[h & pm]) kbuf_push(&buf[h & pm], h);
		}
	}
}

static void extract_long(yko_kbuf_t *buf, int k, int pre, int64_t len, const char *seq) /* count.c:45-60 */
{
	const uint64_t mask = (1ULL << k) - 1, pm = (1ULL << pre) - 1;
	const int shift = k - 1;
	uint64_t x[4] = { 0, 0, 0, 0 };                          /* fwd low/high planes, rev low/high planes */
	int64_t i;
	int run = 0;
	for (i = 0; i < len; ++i) {
		int c = yko_nt4[(uint8_t)seq[i]];
		if (c >= 4) { run = 0; x[0] = x[1] = x[2] = x[3] = 0; continue; }
		x[0] = (x[0] << 1 | (uint64_t)(c & 1)) & mask;
		x[1] = (x[1] << 1 | (uint64_t)(c >> 1)) & mask;
		x[2] = x[2] >> 1 | (uint64_t)(1 - (c & 1)) << shift;
		x[3] = x[3] >> 1 | (uint64_t)(1 - (c >> 1)) << shift;
		if (++run >= k) {
			uint64_t h = yko_hash_long(x);
			if (g_pmask == 0 || g_pmask[h & pm]) kbuf_push(&buf[h & pm], h);
		}
	}
}

void yko_extract(yko_kbuf_t *buf, int k, int pre, int64_t len, const char *seq)
{
	if (k < 32) extract_short(buf, k, pre, len, seq);       /* count.c:121-124 */
	else extract_long(buf, k, pre, len, seq);
}

/* flat variant for tests of the sharded path: every k-mer of a memory image with its stream
 * position (index of its last base), in stream order.  k < 32 only.  Returns the count. */
int64_t yko_extract_pos(int k, const uint8_t *bases, int64_t n, uint64_t *out_hash, uint32_t *out_t)
{
	const uint64_t mask = (1ULL << 2 * k) - 1;
	const int shift = 2 * (k - 1);
	uint64_t fw = 0, rv = 0;
	int64_t i, m = 0;
	int run = 0;
	for (i = 0; i < n; ++i) {
		int c = yko_nt4[bases[i]];
		if (c >= 4) { run = 0; fw = rv = 0; continue; }
		fw = (fw << 2 | (uint64_t)c) & mask;
		rv = rv >> 2 | (uint64_t)(3 - c) << shift;
		if (++run >= k) { out_hash[m] = yko_hash64(fw < rv ? fw : rv, mask); out_t[m++] = (uint32_t)i; }
	}
	return m;
}

/* ------------------------------------------------------------------ FASTA/FASTQ reader
 * Record grammar followed (kseq.h:192-232): skip to a line starting with '>' or '@'; name = up
 * to the first white space, rest of the line ignored; sequence = concatenation of the following
 * lines (a trailing '\r' of a line is dropped once the sequence is longer than one byte) until a
 * line begins with '>', '@' or '+'; after '+': skip that line, read quality lines until at least
 * as many bytes as the sequence; a missing/mismatched quality ends the input (-2, count.c:93). */
typedef struct { char *s; size_t l, m; } fxstr_t;
typedef struct {
	gzFile fp;
	unsigned char *buf;
	int beg, end, eof, last;
	fxstr_t seq, qual, name;
} fx_t;

#define FX_BUF 16384
static int fx_fill(fx_t *f)                                  /* 1 if bytes are available */
{
	if (f->beg < f->end) return 1;
	if (f->eof) return 0;
	f->beg = 0; f->end = gzread(f->fp, f->buf, FX_BUF);
	if (f->end < FX_BUF) f->eof = 1;
	if (f->end <= 0) { f->end = 0; return 0; }
	return 1;
}

static int fx_getc(fx_t *f) { return fx_fill(f) ? f->buf[f->beg++] : -1; }

static void fxstr_add(fxstr_t *d, const unsigned char *src, size_t n)
{
	if (d->l + n + 2 > d->m) { d->m = (d->l + n + 2) * 2; d->s = (char*)realloc(d->s, d->m); }
	memcpy(d->s + d->l, src, n);
	d->l += n;
}

/* consume bytes up to and including a delimiter (line = 1: '\n'; line = 0: any white space),
 * appending them to dst when given.  -1 when already at EOF, else 0.  *dret = delimiter met. */
static int fx_until(fx_t *f, int line, fxstr_t *dst, int *dret)
{
	if (dret) *dret = 0;
	if (f->beg >= f->end && f->eof) return -1;
	while (fx_fill(f)) {
		int i;
		if (line) { for (i = f->beg; i < f->end; ++i) if (f->buf[i] == '\n') break; }
		else { for (i = f->beg; i < f->end; ++i) if (isspace(f->buf[i])) break; }
		if (dst) fxstr_add(dst, f->buf + f->beg, (size_t)(i - f->beg));
		if (i < f->end) { if (dret) *dret = f->buf[i]; f->beg = i + 1; break; }
		f->beg = i + 1;
	}
	if (dst && line && dst->l > 1 && dst->s[dst->l - 1] == '\r') --dst->l;   /* kseq.h:145 */
	return 0;
}

/* returns sequence length, -1 at EOF, -2 on a truncated FASTQ record */
static int64_t fx_read(fx_t *f)
{
	int c, d;
	if (f->last == 0) {
		while ((c = fx_getc(f)) != -1 && c != '>' && c != '@') {}
		if (c == -1) return -1;
		f->last = c;
	}
	f->seq.l = f->qual.l = f->name.l = 0;
	if (fx_until(f, 0, &f->name, &d) < 0) return -1;        /* name */
	if (d != '\n') fx_until(f, 1, 0, 0);                     /* comment */
	while ((c = fx_getc(f)) != -1 && c != '>' && c != '+' && c != '@') {
		unsigned char ch = (unsigned char)c;
		if (c == '\n') continue;
		fxstr_add(&f->seq, &ch, 1);
		fx_until(f, 1, &f->seq, 0);
	}
	if (c == '>' || c == '@') f->last = c;
	if (c != '+') return (int64_t)f->seq.l;                  /* FASTA record */
	while ((c = fx_getc(f)) != -1 && c != '\n') {}
	if (c == -1) return -2;
	while (fx_until(f, 1, &f->qual, 0) >= 0 && f->qual.l < f->seq.l) {}
	f->last = 0;
	if (f->qual.l != f->seq.l) return -2;
	return (int64_t)f->seq.l;
}

/* ------------------------------------------------------------------ chunked serial driver */
static int g_range_lo = -1, g_range_hi = -1;                 /* yko_set_prefix_range: only these sub-tables are counted */
void yko_set_prefix_range(int lo, int hi) { g_range_lo = lo; g_range_hi = hi; }

static uint64_t flush_buffers(yko_ch_t *h, int create_new, yko_kbuf_t *buf)
{
	int p, P = 1 << h->pre;
	uint64_t n_ins = 0;
	for (p = 0; p < P; ++p) {                                /* count.c:133 kt_for over prefixes */
		int64_t off = 0;
		if (g_range_lo >= 0 && (p < g_range_lo || p >= g_range_hi)) { buf[p].n = 0; continue; }   /* the sub-tables are independent of each other */
		/* yak_ch_insert_list takes an int count; feed long buckets in pieces (same put order) */
		while (off < buf[p].n) {
			int64_t w = buf[p].n - off;
			if (w > (1 << 30)) w = 1 << 30;
			n_ins += (uint64_t)yko_ch_insert_list(h, create_new, (int)w, buf[p].a + off);
			off += w;
		}
		buf[p].n = 0;
	}
	return n_ins;
}

static yko_ch_t *table_for(const yko_copt_t *opt, yko_ch_t *h0, int *create_new)
{
	if (h0) {                                               /* count.c:155-157 */
		if (h0->k != opt->k || h0->pre != opt->pre) return 0;
		*create_new = 0;
		return h0;
	}
	*create_new = 1;
	return yko_ch_init(opt->k, opt->pre, opt->bf_n_hash, opt->bf_shift);   /* count.c:160 */
}

yko_ch_t *yko_count_file(const char *fn, const yko_copt_t *opt, yko_ch_t *h0) /* count.c:147-166 */
{
	fx_t f;
	yko_ch_t *h;
	yko_kbuf_t *buf;
	int create_new, p, P;
	int64_t l, sum_len = 0;
	memset(&f, 0, sizeof(f));
	f.fp = (fn == 0 || strcmp(fn, "-") == 0) ? gzdopen(0, "r") : gzopen(fn, "r");
	if (f.fp == 0) return 0;                                 /* count.c:152 */
	f.buf = (unsigned char*)malloc(FX_BUF);
	h = table_for(opt, h0, &create_new);
	if (h == 0) { gzclose(f.fp); free(f.buf); return 0; }
	P = 1 << h->pre;
	buf = (yko_kbuf_t*)calloc(P, sizeof(*buf));
	while ((l = fx_read(&f)) >= 0) {                         /* count.c:93 */
		if (l < opt->k) continue;                            /* count.c:95 */
		yko_extract(buf, opt->k, opt->pre, l, f.seq.s);
		sum_len += l;
		if (sum_len >= opt->chunk_size) {                    /* count.c:106: chunk boundary */
			h->tot += flush_buffers(h, create_new, buf);
			sum_len = 0;
		}
	}
	h->tot += flush_buffers(h, create_new, buf);
	for (p = 0; p < P; ++p) free(buf[p].a);
	free(buf); free(f.seq.s); free(f.qual.s); free(f.buf);
	gzclose(f.fp);
	return h;
}

yko_ch_t *yko_count_mem(const uint8_t *bases, int64_t n, const yko_copt_t *opt, yko_ch_t *h0)
{
	yko_ch_t *h;
	yko_kbuf_t *buf;
	int create_new, p, P;
	int64_t pos = 0, sum_len = 0;
	h = table_for(opt, h0, &create_new);
	if (h == 0) return 0;
	P = 1 << h->pre;
	buf = (yko_kbuf_t*)calloc(P, sizeof(*buf));
	/* a separator resets the rolling k-mer exactly like a record boundary; feed maximal runs of
	 * "anything" in slices so that memory stays bounded */
	while (pos < n) {
		int64_t end = pos + (1 << 24) < n ? pos + (1 << 24) : n;
		/* extend to a separator (or the end) so that no k-mer is cut */
		while (end < n && yko_nt4[bases[end]] < 4) ++end;
		yko_extract(buf, opt->k, opt->pre, end - pos, (const char*)bases + pos);
		sum_len += end - pos;
		pos = end;
		if (sum_len >= opt->chunk_size) { h->tot += flush_buffers(h, create_new, buf); sum_len = 0; }
	}
	h->tot += flush_buffers(h, create_new, buf);
	for (p = 0; p < P; ++p) free(buf[p].a);
	free(buf);
	return h;
}

/* main.c:53-60 */
static yko_ch_t *finish_protocol(yko_ch_t *h, const yko_copt_t *opt,
                                 const char *fn2, const uint8_t *b2, int64_t n2)
{
	if (h == 0) return 0;
	if (opt->bf_shift > 0) {
		yko_ch_destroy_bf(h);
		yko_ch_clear(h);
		h = fn2 ? yko_count_file(fn2, opt, h) : yko_count_mem(b2, n2, opt, h);
		yko_ch_shrink(h, 2, YKO_MAX_COUNT);
	}
	return h;
}

yko_ch_t *yko_count_protocol_mem(const uint8_t *b1, int64_t n1, const uint8_t *b2, int64_t n2,
                                 const yko_copt_t *opt)
{
	yko_ch_t *h = yko_count_mem(b1, n1, opt, 0);
	if (b2 == 0) { b2 = b1; n2 = n1; }
	return finish_protocol(h, opt, 0, b2, n2);
}

yko_ch_t *yko_count_protocol_file(const char *fn1, const char *fn2, const yko_copt_t *opt)
{
	yko_ch_t *h = yko_count_file(fn1, opt, 0);
	return finish_protocol(h, opt, fn2 ? fn2 : fn1, 0, 0);
}


/* test helper: the sequences count.c:93-96 would process (length >= min_len), each followed by '\n' */
int64_t yko_read_image(const char *fn, int min_len, char **out)
{
	fx_t f;
	fxstr_t img = { 0, 0, 0 };
	int64_t len;
	memset(&f, 0, sizeof(f));
	f.fp = (fn && strcmp(fn, "-")) ? gzopen(fn, "r") : gzdopen(0, "r");
	if (!f.fp) return -1;
	f.buf = (unsigned char*)malloc(FX_BUF);
	while ((len = fx_read(&f)) >= 0) {
		if (len < min_len) continue;
		fxstr_add(&img, (const unsigned char*)f.seq.s, (size_t)len);
		fxstr_add(&img, (const unsigned char*)"\n", 1);
	}
	free(f.buf); free(f.seq.s); free(f.qual.s); free(f.name.s);
	gzclose(f.fp);
	*out = img.s ? img.s : (char*)calloc(1, 1);
	return (int64_t)img.l;
}

/* ------------------------------------------------------------------ yak qv counting step (qv.c:34-135)
 * For every sequence of at least min_len bases: t = max(0, count in the table) of each k-mer
 * (canonical, non-ACGT resets the window); tot = number of k-mers, non0 = those present.  Lines the
 * reference prints ("EK" per absent k-mer with -E, "SQ" per sequence with -p) go to `out`.  Sequences
 * with non0 >= tot * min_frac add all their t values to the 1024-bin histogram cnt[]. */
#include <math.h>
void yko_qopt_init(yko_qopt_t *o)                            /* qv.c:137-144 */
{
	memset(o, 0, sizeof(*o));
	o->chunk_size = 1000000000; o->n_threads = 4; o->min_frac = 0.5; o->fpr = 0.00004;
}

int yko_qv(const yko_qopt_t *opt, const char *fn, const yko_ch_t *ch, int64_t *cnt, void *out_)
{
	fx_t f;
	FILE *out = (FILE*)out_;
	int32_t *tv = 0;
	size_t tv_m = 0;
	const int k = ch->k;
	if (k >= 32) return -1;                                   /* qv.c:44 asserts */
	memset(&f, 0, sizeof(f));
	f.fp = (fn && strcmp(fn, "-")) ? gzopen(fn, "r") : gzdopen(0, "r");
	if (!f.fp) return -1;
	f.buf = (unsigned char*)malloc(FX_BUF);
	memset(cnt, 0, sizeof(int64_t) << YKO_COUNTER_BITS);
	const uint64_t mask = (1ULL << 2 * k) - 1;
	const int shift = 2 * (k - 1);
	int64_t len;
	while ((len = fx_read(&f)) >= 0) {
		int64_t i;
		int l = 0, tot = 0, non0 = 0;
		uint64_t x0 = 0, x1 = 0;
		if (f.name.s) f.name.s[f.name.l] = 0;
		if (len < opt->min_len) continue;
		if ((size_t)len > tv_m) { tv_m = (size_t)len * 2; tv = (int32_t*)realloc(tv, tv_m * sizeof(int32_t)); }
		for (i = 0; i < len; ++i) {
			const int c = yko_nt4[(uint8_t)f.seq.s[i]];
			if (c < 4) {
				x0 = (x0 << 2 | (uint64_t)c) & mask;
				x1 = x1 >> 2 | (uint64_t)(3 - c) << shift;
				if (++l >= k) {
					int t = yko_ch_get(ch, yko_hash64(x0 < x1 ? x0 : x1, mask));
					if (t < 0) t = 0;
					if (t > 0) ++non0;
					else if (opt->print_err_kmer && out) fprintf(out, "EK\t%s\t%d\n", f.name.s ? f.name.s : "", (int)(i + 1 - k));
					tv[tot++] = t;
				}
			} else l = 0, x0 = x1 = 0;
		}
		if (opt->print_each && out) {
			double qv = -1.0;
			if (tot > 0) {
				if (non0 > 0) {
					if (tot > non0) { qv = log((double)tot / non0) / k; qv = -4.3429448190325175 * log(qv); }
					else qv = 99.0;
				} else qv = 0.0;
			}
			fprintf(out, "SQ\t%s\t%d\t%d\t%d\t%.2f\n", f.name.s ? f.name.s : "", (int)len, tot, non0, qv);
		}
		if (non0 < tot * opt->min_frac) continue;
		for (i = 0; i < tot; ++i) ++cnt[tv[i]];
	}
	free(tv); free(f.buf); free(f.seq.s); free(f.qual.s); free(f.name.s);
	gzclose(f.fp);
	return 0;
}
