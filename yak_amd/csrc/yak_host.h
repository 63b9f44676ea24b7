/*
 * yak_host.h -- internal: what the three host translation units of the library share (yak_api.cpp: the yak.h surface, dump / restore, qv;
 * yak_reader.cpp: the FASTA / FASTQ reader, its parallel parser and packer, the gzip hooks; yak_multi.cpp: several GPUs behind yak_count()).
 */
#ifndef YAK_HOST_H
#define YAK_HOST_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <assert.h>
#include <zlib.h>
#include <sys/time.h>
#include <sys/resource.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstdarg>
#include <thread>
#include <atomic>
#include <functional>
#include <sys/mman.h>
#include <sys/stat.h>
#include <vector>
#include <map>
#include <algorithm>
#include <string>
#include <mutex>
#include <condition_variable>
#include <cmath>
#include "engine.h"
#include <dlfcn.h>
#include <immintrin.h>
#include "pgz.h"                                            /* parallel inflate of ordinary gzip files */
#include <rccl/rccl.h>                                   /* types and prototypes only: the library is opened when a job asks for several GPUs */
struct yak_ht_t { uint32_t bits, count; uint32_t *used; uint64_t *keys; };
struct yak_ch_ext { yak_ch_t pub; yakamd_ctx *ctx; uint32_t magic; int n_sub; yak_ch_t **sub; };   /* n_sub > 1: sharded over several GPUs, sub[r] owns prefixes [r P / n_sub, (r + 1) P / n_sub) */
#define EXT_MAGIC 0x59414b41u
#define YK_MULTI(e) ((e)->n_sub > 1)
static inline void multi_tot(yak_ch_t *h) { yak_ch_ext *e = (yak_ch_ext*)h; uint64_t t = 0; for (int r = 0; r < e->n_sub; ++r) t += e->sub[r]->tot; h->tot = t; }

double yk_realtime(void);
double yk_cputime(void);

/* ---- the reader (yak_reader.cpp) ---- */
/* ------------------------------------------------------------------------------------------
 * FASTA/FASTQ record reader with the observable behaviour of the reference's parser as driven by
 * count.c:88-110 (record grammar of kseq.h:192-232): header lines start with '>' or '@', the
 * sequence is every following line up to one starting with '>', '@' or '+', a '+' line introduces
 * quality lines covering at least the sequence length; a truncated quality ends the input.
 * ------------------------------------------------------------------------------------------ */
/* What the parallel parser reads: a plain file, or the uncompressed stream of a BGZF file (block gzip: every member carries its
 * compressed size in a 'BC' extra field and holds <= 64 KiB of data, so members can be found without inflating and inflated
 * independently).  A read at any offset inflates just the blocks it touches, on the calling thread -- the parser's threads each
 * read their own segment, so inflation is spread over them by itself.  libdeflate is used when the image has it, else zlib. */
struct ByteSource {
	struct Blk { int64_t foff, uoff; uint32_t csize, usize; };   /* offset of the deflate payload, offset in the uncompressed stream, bytes of both */
	int fd; int64_t size; bool bgzf; std::vector<Blk> blk;
	uint64_t gen;                                                 /* identity of this source for the per-thread block cache (an address can be reused by the next job's source) */
	const unsigned char *map; size_t map_len;                     /* a plain file, mapped: the body of a long FASTA record is stripped of its line ends by several threads straight from here */
	bool pack;                                                    /* the parser threads also pack what they parsed (yakamd_pack_bases_host) */
	bool in_memory, partial;                                      /* bytes in memory (a batch of an inflated gzip stream; map is not ours); more of the stream follows them: a record that touches their end is not finished */
	static uint64_t next_gen() { static uint64_t g = 0; return __atomic_add_fetch(&g, 1, __ATOMIC_RELAXED); }
	ByteSource() : fd(-1), size(0), bgzf(false), gen(next_gen()), map(0), map_len(0), pack(false), in_memory(false), partial(false) {}
	~ByteSource() { if (map && !in_memory) munmap((void*)map, map_len); }
	void set_memory(const unsigned char *p, size_t n, bool more_follows) { fd = -1; bgzf = false; map = p; map_len = n; size = (int64_t)n; in_memory = true; partial = more_follows; }
	ByteSource(const ByteSource&) = delete; ByteSource &operator=(const ByteSource&) = delete;
	void map_plain() {
		if (bgzf || fd < 0 || size <= 0 || map) return;
		void *m = mmap(0, (size_t)size, PROT_READ, MAP_PRIVATE, fd, 0);
		if (m != MAP_FAILED) { map = (const unsigned char*)m; map_len = (size_t)size; (void)madvise(m, map_len, MADV_SEQUENTIAL); }
	}
	typedef void *(*ld_alloc_t)(void); typedef int (*ld_dec_t)(void*, const void*, size_t, void*, size_t, size_t*); typedef void (*ld_free_t)(void*);
	static void ld_api(ld_alloc_t *al, ld_dec_t *de, ld_free_t *fr = 0) {
		static ld_alloc_t a = 0; static ld_dec_t d = 0; static ld_free_t f = 0; static bool tried = false;
		if (!tried) {                                              /* benign race: every thread resolves the same pointers */
			void *l = yk_knob("YAKAMD_NO_LIBDEFLATE", 0) ? 0 : dlopen("libdeflate.so.0", RTLD_NOW);
			if (l) { a = (ld_alloc_t)dlsym(l, "libdeflate_alloc_decompressor"); d = (ld_dec_t)dlsym(l, "libdeflate_deflate_decompress"); f = (ld_free_t)dlsym(l, "libdeflate_free_decompressor"); }
			if (!a || !d) { a = 0; d = 0; f = 0; }
			tried = true;
		}
		*al = a; *de = d; if (fr) *fr = f;
	}
	/* per-thread inflate state, released when the thread ends (the parser starts fresh threads for every window) */
	struct LdState { void *dec; ld_free_t fr; LdState() : dec(0), fr(0) {} ~LdState() { if (dec && fr) fr(dec); } };
	struct ZState { z_stream zs; bool init; ZState() : init(false) { memset(&zs, 0, sizeof(zs)); } ~ZState() { if (init) inflateEnd(&zs); } };
	/* index the members of an open file; false if it is not BGZF from the first byte to the last */
	bool index_bgzf(int f) {
		struct stat sb;
		if (fstat(f, &sb) != 0 || !S_ISREG(sb.st_mode)) return false;
		int64_t off = 0, uoff = 0;
		unsigned char h[18], t[4];
		blk.clear();
		while (off < sb.st_size) {
			if (::pread(f, h, 18, off) != 18) return false;
			if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
			const uint32_t xlen = h[10] | h[11] << 8;
			/* the 'BC' subfield is the first one in every BGZF writer; anything else is not indexed */
			if (xlen < 6 || h[12] != 'B' || h[13] != 'C' || h[14] != 2 || h[15] != 0 || (h[3] & ~4)) return false;
			const uint32_t bsize = (h[16] | h[17] << 8) + 1u;
			if (bsize < 12 + xlen + 8 || off + bsize > sb.st_size) return false;
			if (::pread(f, t, 4, off + bsize - 4) != 4) return false;
			const uint32_t isize = t[0] | t[1] << 8 | t[2] << 16 | (uint32_t)t[3] << 24;
			if (isize > 65536) return false;
			Blk b; b.foff = off + 12 + xlen; b.csize = bsize - 12 - xlen - 8; b.uoff = uoff; b.usize = isize;
			if (isize) blk.push_back(b);
			off += bsize; uoff += isize;
		}
		fd = f; size = uoff; bgzf = true;
		return true;
	}
	bool inflate_block(const Blk &b, unsigned char *out, std::vector<unsigned char> &cbuf) const {
		cbuf.resize(b.csize + 8);
		size_t got = 0;
		while (got < b.csize + 8) { const ssize_t r = ::pread(fd, cbuf.data() + got, b.csize + 8 - got, b.foff + got); if (r <= 0) return false; got += r; }
		ld_alloc_t al; ld_dec_t de; ld_api(&al, &de);
		bool ok = false;
		if (al) {
			static thread_local LdState st;
			if (!st.dec) { st.dec = al(); ld_free_t fr = 0; ld_api(&al, &de, &fr); st.fr = fr; }
			size_t n = 0;
			ok = st.dec && de(st.dec, cbuf.data(), b.csize, out, b.usize, &n) == 0 && n == b.usize;
		} else {
			static thread_local ZState st;
			z_stream &zs = st.zs;
			if (!st.init) { if (inflateInit2(&zs, -15) != Z_OK) return false; st.init = true; } else inflateReset(&zs);
			zs.next_in = cbuf.data(); zs.avail_in = b.csize; zs.next_out = out; zs.avail_out = b.usize;
			ok = inflate(&zs, Z_FINISH) == Z_STREAM_END && zs.avail_out == 0;
		}
		if (!ok) return false;
		const unsigned char *t = cbuf.data() + b.csize;            /* CRC32 of the uncompressed data, as gzread would check it */
		const uint32_t crc = t[0] | t[1] << 8 | t[2] << 16 | (uint32_t)t[3] << 24;
		return (uint32_t)crc32(crc32(0L, Z_NULL, 0), out, b.usize) == crc;
	}
	/* pread(2) semantics on the uncompressed stream; -1 on a corrupt block */
	ssize_t pread_at(void *dst, size_t n, int64_t off) const {
		if (in_memory) { if (off >= size || n == 0) return 0; const size_t take = std::min<size_t>(n, (size_t)(size - off)); memcpy(dst, map + off, take); return (ssize_t)take; }
		if (!bgzf) return ::pread(fd, dst, n, off);
		if (off >= size || n == 0) return 0;
		static thread_local std::vector<unsigned char> ub, cb;
		static thread_local uint64_t who = 0; static thread_local size_t which = (size_t)-1;
		size_t lo = 0, hi = blk.size();                            /* the block that holds `off` */
		while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (blk[mid].uoff <= off) lo = mid; else hi = mid; }
		size_t done = 0;
		for (size_t bi = lo; bi < blk.size() && done < n; ++bi) {
			const Blk &b = blk[bi];
			if (who != gen || which != bi) {
				ub.resize(65536);
				if (!inflate_block(b, ub.data(), cb)) { who = 0; return -1; }
				who = gen; which = bi;
			}
			const size_t skip = (size_t)(off + (int64_t)done - b.uoff), take = std::min<size_t>(b.usize - skip, n - done);
			memcpy((char*)dst + done, ub.data() + skip, take);
			done += take;
		}
		return (ssize_t)done;
	}
};

struct FxReader {
	gzFile fp; int fd; unsigned char *buf; int beg, end, eof, last;
	std::vector<char> seq, name; size_t qlen; int qlast;
	enum { BUF = 1 << 20, NOT_FAST = -3 };
	FxReader() : fp(0), fd(-1), buf(0), beg(0), end(0), eof(0), last(0), qlen(0), qlast(0), mem(false), psrc(0), poff(0), pos0(0) {}
	/* open `fn` (NULL or "-": stdin); a plain (not gzip) regular file is then read with read(2), skipping zlib's copy */
	bool open_file(const char *fn) {
		const bool is_stdin = fn == 0 || strcmp(fn, "-") == 0;
		/* (a pipe is asked for 1 MiB of buffer instead of the usual 64 KiB: 1.9 GB/s instead of 1.1 between `cat` and a reader on this kind of host; the
		 * request fails harmlessly on anything that is no pipe) */
		const int f0 = is_stdin ? 0 : ::open(fn, O_RDONLY);
		if (f0 < 0) return false;
		(void)fcntl(f0, 1031 /* F_SETPIPE_SZ (Linux) */, 1 << 20);
		fp = gzdopen(f0, "r");
		if (fp == 0) { if (!is_stdin) ::close(f0); return false; }
		gzbuffer(fp, 1 << 20);                               /* zlib's default 8 KB means a read() per 8 KB */
		/* plain bytes of a REGULAR file are read directly (no copy through zlib).  Not a FIFO -- `yak count ... <(zcat reads.fq.gz)`, the usual way to
		 * feed gzipped reads, hands over /dev/fd/NN --: gzdirect() has pulled the first buffer out of the pipe already, a second descriptor would go
		 * on behind it and the first megabyte of the stream would be lost (until round 6 it was) */
		struct stat sb_;
		if (!is_stdin && gzdirect(fp) && stat(fn, &sb_) == 0 && S_ISREG(sb_.st_mode)) fd = ::open(fn, O_RDONLY);
		buf = (unsigned char*)malloc(BUF);
		return true;
	}
	void close_file() { if (fd >= 0) ::close(fd); if (fp) gzclose(fp); if (!mem) free(buf); fp = 0; fd = -1; buf = 0; }
	/* positional mode for the parallel parser: read a shared source (plain file or BGZF stream) from offset `from` */
	bool mem; const ByteSource *psrc; int64_t poff, pos0;
	void open_at(const ByteSource *src, int64_t from) { psrc = src; poff = pos0 = from; buf = (unsigned char*)malloc(BUF); beg = end = 0; eof = 0; last = 0; }
	void close_at() { free(buf); buf = 0; }
	/* consume up to the next record marker ('>' or '@', kseq.h:196-199) so that `last` holds it; false at EOF */
	bool seek_marker() {
		if (last != 0) return true;
		int c;
		while ((c = getc()) != -1 && c != '>' && c != '@') {}
		if (c == -1) return false;
		last = c;
		return true;
	}
	int64_t marker_pos() const { return pos0 + (last != 0 ? beg - 1 : end); }   /* file offset of the marker `last` was read from (positional mode) */
	bool fill() {
		if (beg < end) return true;
		if (eof) return false;
		beg = 0;
		if (psrc) {
			pos0 = poff; end = 0;
			while (end < BUF) { const ssize_t r = psrc->pread_at(buf + end, BUF - end, poff); if (r <= 0) break; end += (int)r; poff += r; }
		} else if (fd >= 0) {                                /* read(2) may return short counts before EOF */
			end = 0;
			while (end < BUF) { const ssize_t r = ::read(fd, buf + end, BUF - end); if (r <= 0) break; end += (int)r; }
		} else end = gzread(fp, buf, BUF);
		if (end < BUF) eof = 1;
		if (end <= 0) { end = 0; return false; }
		return true;
	}
	int getc() { return fill() ? buf[beg++] : -1; }
	/* consume through the next delimiter; what: 0 discard, 1 append to seq, 2 count quality bytes, 3 append to name */
	int until(bool line, int what, int *dret) {
		if (dret) *dret = 0;
		if (beg >= end && eof) return -1;
		while (fill()) {
			int i = beg;
			if (line) { const unsigned char *q = (const unsigned char*)memchr(buf + beg, '\n', end - beg); i = q ? (int)(q - buf) : end; }
			else while (i < end && !isspace(buf[i])) ++i;
			if (what == 1) seq.insert(seq.end(), buf + beg, buf + i);
			else if (what == 3) name.insert(name.end(), buf + beg, buf + i);
			else if (what == 2 && i > beg) { qlen += i - beg; qlast = buf[i - 1]; }
			const bool hit = i < end;
			if (hit && dret) *dret = buf[i];
			beg = i + 1;
			if (hit) break;
		}
		if (line && what == 1 && seq.size() > 1 && seq.back() == '\r') seq.pop_back();
		if (line && what == 2 && qlen > 1 && qlast == '\r') { --qlen; qlast = 0; }
		return 0;
	}
	/* Fast path for the record shapes real files are made of -- header line, ONE sequence line, and
	 * either the next record's marker (FASTA) or a '+' line and ONE quality line of the same length
	 * (FASTQ) -- when all of it, plus the byte after it, already sits in the buffer: located with
	 * memchr and appended to `out` (sequence + '\n') straight from the buffer if it has >= min_len
	 * bases.  The reader state it leaves is exactly what next() would leave; anything else (blank or
	 * wrapped lines, CR, a record cut by the buffer end, EOF) returns NOT_FAST without touching the
	 * state, and the caller takes next(). */
	template <class V> int64_t fast(V &out, int64_t min_len) {
		const unsigned char *b = buf;
		int p = beg;
		if (p >= end) return NOT_FAST;
		if (last == 0) { if (b[p] != '@' && b[p] != '>') return NOT_FAST; ++p; }
		const unsigned char *q = (const unsigned char*)memchr(b + p, '\n', end - p);
		if (!q) return NOT_FAST;
		const int s0 = (int)(q - b) + 1;
		if (s0 >= end) return NOT_FAST;
		const int c0 = b[s0];
		if (c0 == '\n' || c0 == '>' || c0 == '+' || c0 == '@') return NOT_FAST;
		q = (const unsigned char*)memchr(b + s0, '\n', end - s0);
		if (!q) return NOT_FAST;
		const int s1 = (int)(q - b), slen = s1 - s0, n0 = s1 + 1;
		if (b[s1 - 1] == '\r' || n0 >= end) return NOT_FAST;
		int nbeg, nlast;
		if (b[n0] == '>' || b[n0] == '@') { nlast = b[n0]; nbeg = n0 + 1; }
		else if (b[n0] == '+') {
			q = (const unsigned char*)memchr(b + n0, '\n', end - n0);
			if (!q) return NOT_FAST;
			const int q0 = (int)(q - b) + 1;
			if ((int64_t)q0 + slen + 1 >= end) return NOT_FAST;
			if (b[q0 + slen] != '\n' || memchr(b + q0, '\n', slen)) return NOT_FAST;
			if (slen > 1 && b[q0 + slen - 1] == '\r') return NOT_FAST;
			nlast = 0; nbeg = q0 + slen + 1;
		} else return NOT_FAST;
		if (slen >= min_len) { out.insert(out.end(), b + s0, b + s1); out.push_back('\n'); }
		beg = nbeg; last = nlast;
		return slen;
	}
	/* The body of a long FASTA record (a chromosome: 1.7 M lines), from a mapped plain file: `n_thr` threads each take a range of the bytes from
	 * `from` on, walk the lines that START in their range -- a line that begins with '>', '@' or '+' ends the body (kseq.h:209) -- and count the
	 * bytes the lines contribute (kseq.h:145: a '\r' before the line end is dropped; the sequence is longer than one byte here); then every thread
	 * copies its lines to their place in `out`.  Returns the offset where the body scan stopped (a line start: the marker line, or the end of
	 * the span / file); out grows by the body's bytes.  `from` must be a line start */
	template <class V> int64_t bulk_body(V &out, int64_t from, int n_thr) {
		const unsigned char *m = psrc->map;
		if (n_thr > 64) n_thr = 64;
		/* a span of 4 MB per thread: a body of 100 MB then keeps every thread busy for several spans (with one span of 1 GB cut into n_thr parts the
		 * first three parts held the whole body and the other threads walked the records behind it for nothing) */
		const int64_t fend = (int64_t)psrc->map_len, span_end = std::min<int64_t>(fend, from + std::max<int64_t>((int64_t)8 << 20, (int64_t)n_thr << 22));
		const int64_t step = (span_end - from + n_thr - 1) / n_thr;
		struct Part { int64_t a, stop, bytes; bool hit; };
		std::vector<Part> part(n_thr);
		int64_t first_hit = INT64_MAX;                                  /* where the body was seen to end: the threads behind it stop counting (their part is not the body's) */
		auto walk = [&](int t, char *dst) {                             /* dst == 0: count; else copy */
			Part &P = part[t];
			int64_t p = P.a;
			const int64_t lim = dst ? P.stop : std::min<int64_t>(span_end, from + (int64_t)(t + 1) * step);
			int64_t nb = 0;
			bool hit = false;
			unsigned n_line = 0;
			while (p < lim) {                                          /* p is a line start */
				const unsigned char c = m[p];
				if (c == '>' || c == '@' || c == '+') {
					hit = true;
					if (!dst) { int64_t cur = __atomic_load_n(&first_hit, __ATOMIC_RELAXED); while (p < cur && !__atomic_compare_exchange_n(&first_hit, &cur, p, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} }
					break;
				}
				if (!dst && (++n_line & 255) == 0 && P.a > __atomic_load_n(&first_hit, __ATOMIC_RELAXED)) break;
				const unsigned char *q = (const unsigned char*)memchr(m + p, '\n', (size_t)(fend - p));
				const int64_t e = q ? (int64_t)(q - m) : fend;
				int64_t len = e - p;
				if (len > 0 && m[e - 1] == '\r') --len;
				if (dst) memcpy(dst + nb, m + p, (size_t)len);
				nb += len;
				p = q ? e + 1 : fend;
			}
			if (!dst) { P.stop = p; P.bytes = nb; P.hit = hit; }
		};
		std::vector<std::thread> th;
		for (int t = 0; t < n_thr; ++t) {
			int64_t a = from + (int64_t)t * step;
			if (t > 0 && a < span_end) {                                /* first line start at or behind the cut */
				const unsigned char *q = (const unsigned char*)memchr(m + a - 1, '\n', (size_t)(fend - (a - 1)));
				a = q ? (int64_t)(q - m) + 1 : fend;
			}
			part[t].a = std::min(a, span_end); part[t].stop = part[t].a; part[t].bytes = 0; part[t].hit = false;
		}
		for (int t = 1; t < n_thr; ++t) th.emplace_back(walk, t, (char*)0);
		walk(0, 0);
		for (auto &x : th) x.join();
		th.clear();
		/* a part whose first line lies beyond its range walked nothing; the body ends at the first marker.  Parts tile the span: part t stops
		 * where part t + 1 starts, unless a marker stopped it */
		int n_use = 0;
		int64_t total = 0, stop = part[0].a;
		std::vector<int64_t> off(n_thr, 0);
		for (int t = 0; t < n_thr; ++t) {
			if (part[t].a != stop) break;                               /* (a long line swallowed this part's range) */
			off[t] = total; total += part[t].bytes; stop = part[t].stop; ++n_use;
			if (part[t].hit) break;
		}
		const size_t at = out.size();
		out.resize(at + (size_t)total);
		char *base = &out[0] + at;
		for (int t = 1; t < n_use; ++t) th.emplace_back(walk, t, base + off[t]);
		if (n_use > 0) walk(0, base + off[0]);
		for (auto &x : th) x.join();
		return stop;
	}
	/* next(), with the sequence appended to `out` (+ '\n') when it has >= min_len bytes, and long FASTA bodies of a mapped file stripped by
	 * bulk_threads threads.  Same return values and reader state as next() */
	template <class V> int64_t next_to(V &out, int64_t min_len, int bulk_threads) {
		if (!(psrc && psrc->map && bulk_threads > 1)) {
			const int64_t l = next();
			if (l >= min_len) { out.insert(out.end(), seq.begin(), seq.end()); out.push_back('\n'); }
			return l;
		}
		int c, d;
		if (last == 0) {
			while ((c = getc()) != -1 && c != '>' && c != '@') {}
			if (c == -1) return -1;
			last = c;
		}
		seq.clear(); name.clear(); qlen = 0; qlast = 0;
		if (until(false, 3, &d) < 0) return -1;
		if (d != '\n') until(true, 0, 0);
		const size_t at0 = out.size();
		int64_t bulked = 0;
		while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {
			if (c == '\n') continue;
			seq.push_back((char)c);
			until(true, 1, 0);
			if (seq.size() >= ((size_t)1 << 20)) {                       /* a long body: what is read so far goes out, the rest in parallel from the map */
				out.insert(out.end(), seq.begin(), seq.end());
				bulked += (int64_t)seq.size();
				seq.clear();
				int64_t p = pos0 + beg;                                 /* the reader stands at a line start (or at the end of the file) */
				for (;;) {
					const size_t before = out.size();
					/* room for the spans to come in one step (a vector that grows span by span copies the whole body again and again, on one thread:
					 * a quarter of the time of a 100 Mb record); address space only until it is written */
					const size_t ahead = (size_t)std::min<int64_t>((int64_t)psrc->map_len - p, (int64_t)1 << 30) + ((size_t)1 << 20);
					if (out.capacity() - before < std::min<size_t>(ahead, (size_t)bulk_threads << 22)) { try { out.reserve(before + ahead); } catch (const std::bad_alloc&) {} }
					const int64_t q = bulk_body(out, p, bulk_threads);
					bulked += (int64_t)(out.size() - before);
					const bool more = q > p && q < (int64_t)psrc->map_len && psrc->map[q] != '>' && psrc->map[q] != '@' && psrc->map[q] != '+';   /* the span ended before the body did */
					p = q;
					if (!more) break;
				}
				beg = end = 0; eof = 0; poff = p;                       /* the buffered reader goes on from there */
				seq.push_back('x'); seq.push_back('x');                 /* (kseq.h:145 looks at the sequence's length: "more than one byte" stays true) */
			}
		}
		const int64_t slen = bulked ? bulked + (int64_t)seq.size() - 2 : (int64_t)seq.size();
		if (bulked) { out.insert(out.end(), seq.begin() + 2, seq.end()); }
		else if ((int64_t)seq.size() >= min_len) out.insert(out.end(), seq.begin(), seq.end());
		if (c == '>' || c == '@') last = c;
		int64_t ret = slen;
		if (c == '+') {
			while ((c = getc()) != -1 && c != '\n') {}
			if (c == -1) ret = -2;
			else {
				while (until(true, 2, 0) >= 0 && (int64_t)qlen < slen) {}
				last = 0;
				if ((int64_t)qlen != slen) ret = -2;
			}
		}
		if (ret >= min_len) out.push_back('\n'); else out.resize(at0);
		return ret;
	}
	int64_t next() {
		int c, d;
		if (last == 0) {
			while ((c = getc()) != -1 && c != '>' && c != '@') {}
			if (c == -1) return -1;
			last = c;
		}
		seq.clear(); name.clear(); qlen = 0; qlast = 0;
		if (until(false, 3, &d) < 0) return -1;
		if (d != '\n') until(true, 0, 0);
		while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {
			if (c == '\n') continue;
			seq.push_back((char)c);
			until(true, 1, 0);
		}
		if (c == '>' || c == '@') last = c;
		if (c != '+') return (int64_t)seq.size();
		while ((c = getc()) != -1 && c != '\n') {}
		if (c == -1) return -2;
		while (until(true, 2, 0) >= 0 && qlen < seq.size()) {}
		last = 0;
		return qlen == seq.size() ? (int64_t)seq.size() : -2;
	}
};
/* the accepted segments of a window as packed pieces, in stream order (yakamd_feed_packed_pieces_host lays them out on the device: one feed) */
struct WinPack { std::vector<const void*> codes, valid; std::vector<int64_t> n_words; int64_t n_pos, n_seq; WinPack() : n_pos(0), n_seq(0) {} };
/* what takes the parsed pieces, in stream order: the base image (sequences, each followed by '\n'), its bytes, its sequences, and -- when the
 * source asked for it (ByteSource::pack) -- no ASCII image but the packed pieces of a whole window (n = its stream positions), else 0 */
typedef std::function<bool(const char*, size_t, int64_t, const WinPack*)> ImgSink;
extern std::atomic<double> g_t_parse_windows, g_t_first_window;   /* YAKAMD_VERBOSE: wall time of the window parses (they overlap the sink), of the first one; statistics of the LAST call when several run at once */
/* memory handed back behind the caller's back (some GB of images, the inflater's buffers: ~0.1 s of munmap): on a thread that the library joins before it goes --
 * at exit() or dlclose() -- and not on a detached one, which could still be inside free() while the process tears down or the code is unmapped */
void yk_reap_later(std::function<void()> f);
int parse_threads(int n_thread);
bool parallel_source(const char *fn, const FxReader &fx, int n_thr, int64_t min_size, ByteSource *src, int *own_fd);
bool parse_parallel(const ByteSource *fd, int min_len, int n_thr, const ImgSink &sink, int64_t *stopped_at = 0, bool *stream_ended = 0);
bool gz_source(const char *fn, const FxReader &fx, int n_thr, pgz::Reader *z);
bool parse_gz(pgz::Reader *z, int min_len, int n_thr, const ImgSink &sink, bool pack = false);

/* ---- several GPUs behind yak_count() (yak_multi.cpp) ---- */
bool env_fast_default();
int auto_sweeps(const yak_copt_t *opt, const char *fn);
int multi_gpus(const yak_copt_t *opt, std::vector<int> *dev, const char *fn = 0);
yak_ch_t *yak_count_multi(const char *fn, const yak_copt_t *opt, yak_ch_t *h0, int N, const std::vector<int> &dev);
#endif
