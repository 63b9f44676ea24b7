/*
 * yak_reader.cpp -- the host input stage (reference count.c:88-110, kseq.h): parallel parsing of plain and block-gzipped files, the
 * packer of the base image, the reader of ordinary gzip files (pgz.h).  The record reader itself (FxReader) and the byte source are in yak_host.h.
 */
#include "yak_host.h"
#include <memory>


/* ------------------------------------------------------------------------------------------
 * Parallel parsing of a plain (uncompressed, mapped) file.  A window of the file is cut into one
 * segment per thread.  Segment 0 starts at a verified record boundary; the others start at a GUESS
 * (first line after the cut that begins with '>' or, for '@', whose third line begins with '+').
 * Every thread parses records with the ordinary reader until the next record would start at or
 * beyond its segment's end and reports where that is.  A segment's output is accepted only if the
 * previous accepted segment stopped exactly at its start -- so the accepted stream is, by induction,
 * what the single reader would have produced; the next window starts where the last accepted
 * segment stopped.  A wrong guess costs time, never correctness.
 * ------------------------------------------------------------------------------------------ */
static int64_t env_threads_window() { const int64_t w = yk_knob("YAKAMD_PARSE_WINDOW", 0); return w > 0 ? w : 0; }
int parse_threads(int n_thread)
{
	const char *e = getenv("YAKAMD_PARSE_THREADS");
	int n = e ? atoi(e) : n_thread;
	const int hw = (int)std::thread::hardware_concurrency();
	if (hw > 0 && n > hw) n = hw;
	return n < 1 ? 1 : n > 32 ? 32 : n;
}

/* The parsed base images' buffers (reused from window to window; pageable: page-locking them cost more than the runtime's staged copies of
 * pageable memory -- CLI run 1.44 s against 2.2 s -- and the multi-GPU reader stages through its own two pinned buffers) */
template <class T> struct PinAlloc {
	typedef T value_type;
	PinAlloc() {}
	template <class U> PinAlloc(const PinAlloc<U>&) {}
	T *allocate(size_t n) {
		void *p = malloc(n * sizeof(T) + 16);
		if (!p) throw std::bad_alloc();
		return (T*)p;
	}
	void deallocate(T *p, size_t) { free((void*)p); }
	template <class U> void construct(U*) {}                        /* resize() leaves new bytes alone: they are written right away (no zero fill of a 100 MB sequence) */
	template <class U, class A0> void construct(U *p, const A0 &a) { ::new ((void*)p) U(a); }
	template <class U> bool operator==(const PinAlloc<U>&) const { return true; }
	template <class U> bool operator!=(const PinAlloc<U>&) const { return false; }
};
typedef std::vector<char, PinAlloc<char> > PinVec;

/* ---- the base image packed on the host (include/yak_amd.h: yakamd_feed_packed_dev's format): 2-bit codes, 16 bases per 32-bit word, and one
 * validity bit per base, by the table the kernels use (seq_nt4_table, reference yak.h / count.c:28-31: ACGT, acgt, U, u and the bytes 0..3 are
 * bases, everything else -- N, the '\n' between two records -- is not) ---- */
static const uint8_t yk_nt4[256] = {
#define R4(v) v, v, v, v
#define R16(v) R4(v), R4(v), R4(v), R4(v)
	0, 1, 2, 3, R4(4), R4(4), R4(4),
	R16(4), R16(4), R16(4),
	4, 0, 4, 1, 4, 4, 4, 2, R4(4), R4(4),
	4, 4, 4, 4, 3, 3, 4, 4, R4(4), R4(4),
	4, 0, 4, 1, 4, 4, 4, 2, R4(4), R4(4),
	4, 4, 4, 4, 3, 3, 4, 4, R4(4), R4(4),
	R16(4), R16(4), R16(4), R16(4), R16(4), R16(4), R16(4), R16(4)
#undef R16
#undef R4
};
static inline void pack32_scalar(const uint8_t *a, int64_t left, uint32_t *c0, uint32_t *c1, uint32_t *v)
{
	uint32_t x0 = 0, x1 = 0, m = 0;
	const int n = left < 32 ? (int)left : 32;
	for (int j = 0; j < n; ++j) {
		const uint32_t c = yk_nt4[a[j]];
		if (c < 4) { m |= 1u << j; if (j < 16) x0 |= c << (2 * j); else x1 |= c << (2 * (j - 16)); }
	}
	*c0 = x0; *c1 = x1; *v = m;
}
/* 32 bases per step with AVX2 + BMI2: A / C / G / T of either case by four compares (the validity word is their movemask), the code of such a
 * byte is bits 1..2 of it with bit 1 flipped when bit 2 is set (A 0x41 -> 0, C 0x43 -> 1, G 0x47 -> 2, T 0x54 -> 3), gathered by pext; a group
 * that holds one of the rare other bases (U, u, a raw 0..3) goes through the table */
__attribute__((target("avx2,bmi2")))
static void pack_words_avx2(const uint8_t *a, int64_t n_words, uint32_t *codes, uint32_t *valid)
{
	const __m256i up = _mm256_set1_epi8((char)0xDF), A = _mm256_set1_epi8('A'), C = _mm256_set1_epi8('C'), G = _mm256_set1_epi8('G'), T = _mm256_set1_epi8('T'),
	              U = _mm256_set1_epi8('U'), four = _mm256_set1_epi8(4), b4 = _mm256_set1_epi8(0x04);
	for (int64_t w = 0; w < n_words; ++w, a += 32) {
		const __m256i x = _mm256_loadu_si256((const __m256i*)a), u = _mm256_and_si256(x, up);
		const __m256i acgt = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, A), _mm256_cmpeq_epi8(u, C)), _mm256_or_si256(_mm256_cmpeq_epi8(u, G), _mm256_cmpeq_epi8(u, T)));
		const __m256i rare = _mm256_or_si256(_mm256_cmpeq_epi8(u, U), _mm256_cmpeq_epi8(_mm256_min_epu8(x, four), x) /* x <= 4 */);
		const __m256i rare4 = _mm256_andnot_si256(_mm256_cmpeq_epi8(x, four), rare);   /* x < 4, or U / u */
		if (_mm256_movemask_epi8(rare4)) { pack32_scalar(a, 32, &codes[2 * w], &codes[2 * w + 1], &valid[w]); continue; }
		const uint32_t m = (uint32_t)_mm256_movemask_epi8(acgt);
		const __m256i y = _mm256_xor_si256(x, _mm256_srli_epi16(_mm256_and_si256(x, b4), 1));
		uint64_t q[4];
		_mm256_storeu_si256((__m256i*)q, y);
		const uint64_t sel = 0x0606060606060606ull;
		const uint64_t code = _pext_u64(q[0], sel) | _pext_u64(q[1], sel) << 16 | _pext_u64(q[2], sel) << 32 | _pext_u64(q[3], sel) << 48;
		const uint64_t keep = _pdep_u64((uint64_t)m, 0x5555555555555555ull) * 3;
		const uint64_t cv = code & keep;
		codes[2 * w] = (uint32_t)cv; codes[2 * w + 1] = (uint32_t)(cv >> 32); valid[w] = m;
	}
}
/* n bases -> (n + 31) / 32 words of validity bits and twice as many of codes; the bits behind base n - 1 in the last words are zero */
static void pack_into(const uint8_t *a, int64_t n, uint32_t *codes, uint32_t *valid)
{
	const int64_t nw = (n + 31) / 32, whole = n / 32;
	static const bool wide = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !yk_knob("YAKAMD_NO_AVX2", 0);
	int64_t w = 0;
	if (wide) { pack_words_avx2(a, whole, codes, valid); w = whole; }
	for (; w < nw; ++w) pack32_scalar(a + 32 * w, n - 32 * w, &codes[2 * w], &codes[2 * w + 1], &valid[w]);
}
extern "C" void yakamd_pack_bases_host(const void *ascii, int64_t n, void *h_packed)
{
	if (n <= 0) return;
	const int64_t nw = (n + 31) / 32;
	uint32_t *codes = (uint32_t*)h_packed, *valid = (uint32_t*)((char*)h_packed + ((nw * 8 + 15) & ~(int64_t)15));
	pack_into((const uint8_t*)ascii, n, codes, valid);
	const int64_t pad = (((nw * 8 + 15) & ~(int64_t)15) - nw * 8) / 4;
	for (int64_t i = 0; i < pad; ++i) codes[2 * nw + i] = 0;
}

typedef std::vector<uint32_t, PinAlloc<uint32_t> > PinWords;
/* a parsed segment: its base image -- or, when the source asks for packed pieces, that image packed as it grows (the ASCII bytes then only pass
 * through a buffer of ~1 MB that stays in the thread's cache): code words, validity words, 32 stream positions per validity word.  The image
 * of a segment is taken to end at a multiple of 32 positions: the up to 31 positions behind it are no bases (stream positions only order the
 * k-mers; a few of them unused change nothing), so a window's segments can be laid one behind the other word by word */
struct ParSeg { int64_t start, end, stop; PinVec img; PinWords codes, valid; int64_t n_seq, sum_len; bool hard_end; };


static int64_t guess_record_start(const ByteSource *src, int64_t from, int64_t limit)
{
	std::vector<unsigned char> tmp((size_t)(limit - from));
	int64_t got = 0;
	while (got < (int64_t)tmp.size()) { const ssize_t r = src->pread_at(tmp.data() + got, tmp.size() - got, from + got); if (r <= 0) break; got += r; }
	const unsigned char *base = tmp.data(), *p = base, *e = base + got;
	p = (const unsigned char*)memchr(p, '\n', e - p);
	if (!p) return -1;
	for (++p; p < e; ) {
		const unsigned char *l1 = (const unsigned char*)memchr(p, '\n', e - p);
		if (*p == '>') return from + (p - base);
		if (*p == '@' && l1) {
			const unsigned char *l2 = l1 + 1 < e ? (const unsigned char*)memchr(l1 + 1, '\n', e - (l1 + 1)) : 0;
			if (l2 && l2 + 1 < e && l2[1] == '+') return from + (p - base);
		}
		if (!l1) return -1;
		p = l1 + 1;
	}
	return -1;
}

static void parse_segment(const ByteSource *src, int64_t file_end, ParSeg *sg, int min_len, int bulk_threads)
{
	FxReader r;
	r.open_at(src, sg->start);
	sg->n_seq = sg->sum_len = 0; sg->hard_end = false;
	sg->img.clear(); sg->codes.clear(); sg->valid.clear();
	const bool pack = src->pack;
	if (!pack && sg->img.capacity() < (size_t)(sg->end - sg->start)) sg->img.reserve((size_t)(sg->end - sg->start) + (1 << 16));   /* the sequences are a part of the segment's bytes */
	if (pack) { const size_t w = (size_t)(sg->end - sg->start) / 32 + 64; if (sg->valid.capacity() < w) { sg->valid.reserve(w); sg->codes.reserve(2 * w); } }
	auto flush = [&](bool all) {                                  /* whole words of the staged bases go to the packed image; at the end the rest too, padded */
		const size_t n = all ? sg->img.size() : sg->img.size() & ~(size_t)31;
		if (n == 0) return;
		const size_t w0 = sg->valid.size(), nw = (n + 31) / 32;
		sg->valid.resize(w0 + nw); sg->codes.resize(2 * (w0 + nw));
		pack_into((const uint8_t*)sg->img.data(), (int64_t)n, &sg->codes[2 * w0], &sg->valid[w0]);
		const size_t rest = sg->img.size() - n;
		if (rest) memmove(&sg->img[0], &sg->img[n], rest);
		sg->img.resize(rest);
	};
	int64_t l;
	for (;;) {
		if (pack && sg->img.size() >= ((size_t)1 << 20)) flush(false);
		if (!r.seek_marker()) { sg->stop = file_end; sg->hard_end = !src->partial; break; }
		const int64_t mp = r.marker_pos();
		if (mp >= sg->end) { sg->stop = mp; break; }
		const size_t img0 = sg->img.size();
		if ((l = r.fast(sg->img, min_len)) == FxReader::NOT_FAST) l = r.next_to(sg->img, min_len, bulk_threads);
		/* more of the stream follows these bytes and the reader has used them up: the record may go on there (`last` still holds the
		 * marker it started with, kseq.h:186-190, so it cannot tell) -- it is left, from its marker on, for the next batch */
		if (src->partial && !r.fill()) { sg->img.resize(img0); sg->stop = mp; break; }
		if (l < 0) { sg->stop = file_end; sg->hard_end = true; break; }   /* EOF inside a record, or a truncated FASTQ record: the stream ends (count.c:93) */
		if (l >= min_len) { ++sg->n_seq; sg->sum_len += l; }
	}
	r.close_at();
	if (pack) flush(true);
}

/* one window: cut [pos, wend) into segments, parse them on n_thr threads, accept the verified prefix.  Returns the
 * number of accepted segments; *next = where the following window starts; *done = the stream has ended */
static int parse_window(const ByteSource *fd, int64_t size, int64_t pos, int64_t WIN, int min_len, int n_thr, std::vector<ParSeg> &seg, int64_t *next, bool *done, WinPack *wp)
{
	const int64_t wend = std::min(size, pos + WIN), step = (wend - pos + n_thr - 1) / n_thr;
	int n_seg = 0;
	for (int i = 0; i < n_thr; ++i) {
		const int64_t cut = pos + i * step;
		if (cut >= wend) break;
		const int64_t st = i == 0 ? pos : guess_record_start(fd, cut, std::min(size, cut + ((int64_t)1 << 18)));
		if (i && (st < 0 || st >= wend)) continue;
		if (n_seg && st <= seg[n_seg - 1].start) continue;
		seg[n_seg].start = st; ++n_seg;
	}
	for (int i = 0; i < n_seg; ++i) seg[i].end = i + 1 < n_seg ? seg[i + 1].start : wend;
	std::vector<std::thread> th;
	/* few segments (long records: a cut finds no record start nearby): their threads' share of the parser threads strips the long bodies */
	const int bulk_threads = std::max(1, std::min(n_thr, (int)std::thread::hardware_concurrency()) / std::max(1, n_seg));
	for (int i = 1; i < n_seg; ++i) th.emplace_back(parse_segment, fd, size, &seg[i], min_len, bulk_threads);
	parse_segment(fd, size, &seg[0], min_len, bulk_threads);
	for (auto &t : th) t.join();
	int64_t at = pos;
	int n_ok = 0;
	for (int i = 0; i < n_seg; ++i) {
		if (seg[i].start != at) break;                           /* wrong guess: the rest of the window is parsed again */
		++n_ok;
		at = seg[i].stop;
		if (seg[i].hard_end) { *done = true; break; }
	}
	*next = at;
	if (fd->pack) {
		wp->codes.clear(); wp->valid.clear(); wp->n_words.clear(); wp->n_pos = wp->n_seq = 0;
		for (int i = 0; i < n_ok; ++i) {
			wp->n_seq += seg[i].n_seq;
			if (seg[i].valid.empty()) continue;
			wp->codes.push_back(seg[i].codes.data()); wp->valid.push_back(seg[i].valid.data()); wp->n_words.push_back((int64_t)seg[i].valid.size());
			wp->n_pos += 32 * (int64_t)seg[i].valid.size();
		}
	}
	return n_ok;
}

/* the source the parallel parser can take for `fn`, if any: a plain regular file (fx.fd) or a BGZF file, larger than min_size.
 * *own_fd (>= 0) is a descriptor the caller closes afterwards */
bool parallel_source(const char *fn, const FxReader &fx, int n_thr, int64_t min_size, ByteSource *src, int *own_fd)
{
	*own_fd = -1;
	if (n_thr <= 1) return false;
	struct stat sb;
	if (fx.fd >= 0) {
		if (fstat(fx.fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size <= min_size) return false;
		src->fd = fx.fd; src->size = sb.st_size; src->bgzf = false;
		src->map_plain();
		return true;
	}
	if (fn == 0 || strcmp(fn, "-") == 0) return false;
	if (stat(fn, &sb) != 0 || !S_ISREG(sb.st_mode)) return false;   /* (never open() a FIFO a second time: with its writer gone -- a short input, all of it in the pipe already -- that open blocks for ever) */
	const int f = ::open(fn, O_RDONLY);
	if (f < 0) return false;
	if (!src->index_bgzf(f) || src->size <= min_size) { ::close(f); src->fd = -1; src->bgzf = false; return false; }
	*own_fd = f;
	return true;
}

/* calls sink(image bytes, n_bytes, n_seq) for consecutive pieces of the input, in order; false if sink failed.
 * Two sets of segment buffers: while the sink consumes one window (copy to the device + kernels), the parser
 * threads already work on the next one. */
namespace {
struct Reaper {
	struct Job { std::thread t; std::shared_ptr<std::atomic<bool>> done; };
	std::mutex mu; std::vector<Job> jobs;
	void later(std::function<void()> f) {
		std::lock_guard<std::mutex> g(mu);
		for (size_t i = 0; i < jobs.size(); ) if (jobs[i].done->load(std::memory_order_acquire)) { jobs[i].t.join(); jobs[i] = std::move(jobs.back()); jobs.pop_back(); } else ++i;
		auto done = std::make_shared<std::atomic<bool>>(false);
		jobs.push_back(Job{ std::thread([f, done]() { f(); done->store(true, std::memory_order_release); }), done });
	}
	~Reaper() { for (Job &j : jobs) if (j.t.joinable()) j.t.join(); }
};
Reaper g_reaper;                                                /* destroyed -- its threads joined -- when the process exits or the library is closed */
}
void yk_reap_later(std::function<void()> f) { g_reaper.later(std::move(f)); }

std::atomic<double> g_t_parse_windows{0}, g_t_first_window{0};    /* YAKAMD_VERBOSE: wall time of the window parses (they overlap the sink), of the first one */
/* A parser thread fills a ring of window sets while the caller's thread hands the finished windows to the sink, in order: two sets for
 * ASCII pieces (the sink copies a window to the device while the next one is parsed), four when the windows are packed -- a new table's
 * first feed waits ~0.25 s for the runtime to come up, time in which the parser gets through 2 GB of file instead of standing still */
bool parse_parallel(const ByteSource *fd, int min_len, int n_thr, const ImgSink &sink, int64_t *stopped_at, bool *stream_ended)
{
	if (stopped_at) *stopped_at = 0;
	if (stream_ended) *stream_ended = false;
	const int64_t size = fd->size;
	if (size <= 0) return true;
	/* the windows grow from 128 MiB to 1 GiB: the device has its first piece after an eighth of the time a full window takes to parse */
	const int64_t win_set = env_threads_window();
	struct WinSet { std::vector<ParSeg> seg; WinPack wp; int n_ok; int64_t next; bool done; };
	const int NSET = fd->pack ? 4 : 2;
	std::vector<WinSet> *ring_p = new std::vector<WinSet>(NSET);
	std::vector<WinSet> &ring = *ring_p;
	for (auto &w : ring) { w.seg.resize(n_thr); w.n_ok = 0; w.next = 0; w.done = false; }
	std::mutex mu; std::condition_variable cv;
	int produced = 0, consumed = 0;
	bool prod_end = false, abort = false;
	std::thread producer([&]() {
		int64_t pos = 0;
		for (int k = 0; ; ++k) {
			{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return abort || k - consumed < NSET; }); if (abort) break; }
			WinSet &w = ring[k % NSET];
			const int64_t WIN = win_set ? win_set : std::min<int64_t>((int64_t)1 << 30, (int64_t)128 << 20 << std::min(k, 3));
			const double t = yk_realtime();
			w.done = false;
			w.n_ok = parse_window(fd, size, pos, WIN, min_len, n_thr, w.seg, &w.next, &w.done, &w.wp);
			const double dt = yk_realtime() - t;
			g_t_parse_windows.store(g_t_parse_windows.load(std::memory_order_relaxed) + dt, std::memory_order_relaxed); if (k == 0) g_t_first_window.store(dt, std::memory_order_relaxed);
			/* (a partial source: a window that gets nowhere stands at a record that wants the bytes still to come) */
			const bool more = !w.done && w.next < size && !(fd->partial && w.next == pos);
			pos = w.next;
			{ std::lock_guard<std::mutex> lk(mu); ++produced; if (!more) prod_end = true; }
			cv.notify_all();
			if (!more) break;
		}
		{ std::lock_guard<std::mutex> lk(mu); prod_end = true; }
		cv.notify_all();
	});
	bool ok = true;
	for (int k = 0; ; ++k) {
		{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return produced > k || prod_end; }); if (produced <= k) break; }
		WinSet &w = ring[k % NSET];
		if (stopped_at) *stopped_at = w.next;
		if (stream_ended) *stream_ended = w.done;
		if (fd->pack) { if (w.n_ok > 0 && (w.wp.n_pos > 0 || w.wp.n_seq > 0)) ok = sink(0, (size_t)w.wp.n_pos, w.wp.n_seq, &w.wp); }
		else for (int i = 0; i < w.n_ok && ok; ++i) if (!w.seg[i].img.empty() || w.seg[i].n_seq > 0) ok = sink(w.seg[i].img.data(), w.seg[i].img.size(), w.seg[i].n_seq, 0);
		{ std::lock_guard<std::mutex> lk(mu); ++consumed; if (!ok) abort = true; }
		cv.notify_all();
		if (!ok) break;
	}
	producer.join();
	yk_reap_later([ring_p]() { delete ring_p; });                 /* (giving some GB of images back to the system takes ~0.1 s: not in the caller's way) */
	return ok;
}

/* an ordinary gzip file: batches of it are inflated by several threads (pgz.h) while the batch before is parsed, by the same window
 * parser, from memory; the record a batch ends in is carried to the front of the next one */
bool gz_source(const char *fn, const FxReader &fx, int n_thr, pgz::Reader *z)
{
	if (n_thr <= 1 || fx.fd >= 0 || fn == 0 || strcmp(fn, "-") == 0 || yk_knob("YAKAMD_NO_PGZ", 0)) return false;
	pgz::tune().no_simd = yk_knob("YAKAMD_NO_AVX2", 0) != 0;
	return z->open(fn, n_thr);
}
bool parse_gz(pgz::Reader *z, int min_len, int n_thr, const ImgSink &sink, bool pack)
{
	size_t keep = 0;
	for (bool last = false; !last; ) {
		uint8_t *p = 0; size_t n = 0;
		if (!z->next(keep, &p, &n, &last)) { yk_set_error("%s", z->why.c_str()); return false; }
		ByteSource src;
		src.set_memory(p, n, !last);
		src.pack = pack;
		int64_t stop = 0; bool ended = false;
		if (!parse_parallel(&src, min_len, n_thr, sink, &stop, &ended)) return false;
		if (ended) break;                                         /* a truncated record ended the stream (count.c:93) */
		keep = (size_t)stop;
	}
	if (getenv("YAKAMD_VERBOSE")) fprintf(stderr, "[yak_amd] gzip: %d threads inflated %lu chunks from a searched block start (%lu searched starts not used, %.1f MB decoded by the stitch)\n",
	                                      z->n_thr, (unsigned long)z->n_search_ok, (unsigned long)z->n_search_bad, z->n_gap_bits / 8e6);
	return true;
}

