/* Parallel inflate of an ordinary gzip file (one member, or several concatenated; a BGZF file is taken by its own
 * block index before this is tried).  Replaces, for yak_count()'s reader, the gzread() behind kseq
 * (reference kseq.h:80-96 ks_getc/ks_getuntil -> __read = gzread; count.c:88-110 is the loop that drains it):
 * the bytes delivered are the bytes gzread() delivers, the CRC32 / ISIZE of every member are checked.
 *
 * DEFLATE (RFC 1951) has no block index, so a batch of the compressed file is cut into one chunk per thread and
 *   chunk 0   starts at a known bit with a known 32 KiB window: decoded to bytes;
 *   chunk c>0 SEARCHES the first bit at or behind its cut where a non-final dynamic-Huffman block starts (a header
 *             with complete code-length / literal / distance codes whose block decodes to text and is followed by a
 *             plausible header), then decodes from there with an UNKNOWN window: 16-bit symbols, a literal as
 *             itself, a byte that would come out of the window as 256 + its window position;
 * every chunk decodes whole blocks until one would start behind the next cut.  A sequential stitch then walks the
 * chunks: chunk c is accepted only if the decode before it ended exactly on the bit c started from (else the bits
 * in between -- or all of c, after a false start -- are decoded again, exactly, from the known state); its window
 * is resolved from the window before it (32 Ki table look-ups), and all chunks are then translated to bytes and
 * CRC'd in parallel.  A wrong guess costs time, never bytes.
 *
 * Errors: a truncated stream delivers every complete symbol and ends, as gzread() does; invalid DEFLATE data, a bad
 * CRC32 or ISIZE fail the reader (gzread() would fail too, after having delivered a prefix). */
#ifndef YAKAMD_PGZ_H
#define YAKAMD_PGZ_H

#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <zlib.h>                                            /* crc32, crc32_combine */
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <vector>
#include <string>
#include <thread>
#include <mutex>
#include <algorithm>
#include <immintrin.h>

namespace pgz {

enum { WSIZE = 32768, LT_BITS = 10, DT_BITS = 8, LT_SIZE = (1 << LT_BITS) + 288 * 32, DT_SIZE = (1 << DT_BITS) + 32 * 128 };   /* (zlib closes a block every 16 K symbols: the tables are built thousands of times per 100 MB -- a first level of 2^10 entries costs half of what 2^11 does) */
enum { E_INVALID = 0x8000, E_LINK = 0x4000, E_EOB = 0x2000, E_BASE = 0x1000 };   /* entry = value << 16 | kind | extra or sub-table bits << 8 | bits to drop */
enum { D_OK = 0, D_ROOM = 1, D_DATA = -1, D_TRUNC = -2, D_TEXT = -3 };

struct Tune { size_t chunk, min_size, front; bool no_simd; Tune() : chunk((size_t)1 << 20), min_size((size_t)4 << 20), front((size_t)64 << 20), no_simd(false) {} };   /* compressed bytes per thread and batch; smallest file taken; room kept in front of a batch for what the consumer carries over; no CLMUL / SSE paths (tests) */
static Tune &tune() { static Tune t; return t; }

static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

struct BitIn {
	const uint8_t *base, *p, *end;
	uint64_t bb; int bc;                                       /* bc < 0: more bits were taken than the input holds */
	void init(const uint8_t *b, size_t n, uint64_t bit) {
		base = b; end = b + n; p = b + std::min<uint64_t>(bit >> 3, n); bb = 0; bc = 0;
		refill(); drop((int)(bit & 7));
	}
	inline void refill() {
		if (p + 8 <= end) { bb |= ld64(p) << bc; p += (63 - bc) >> 3; bc |= 56; }
		else while (bc <= 56 && p < end) { bb |= (uint64_t)*p++ << bc; bc += 8; }
	}
	inline void drop(int n) { bb >>= n; bc -= n; }
	inline uint32_t take(int n) { const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1)); drop(n); return v; }
	uint64_t bit_pos() const { return (uint64_t)(p - base) * 8 - (uint64_t)bc; }   /* valid while bc >= 0 */
};

static const uint16_t LEN_BASE[29] = { 3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258 };
static const uint8_t LEN_XTRA[29] = { 0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0 };
static const uint16_t DIST_BASE[30] = { 1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577 };
static const uint8_t DIST_XTRA[30] = { 0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13 };

/* canonical Huffman code of `n` lengths (<= 15 bits) -> a table indexed by the next TB bits of the stream (codes are sent
 * most significant bit first, so by the reversed code), longer codes through a second-level table.  kind: 0 literal/length,
 * 1 distance, 2 code lengths.  Returns 0, or -1 for the sets zlib's inflate_table() rejects: over-subscribed, or
 * incomplete (an incomplete set is allowed only as ONE code of one bit in a literal/length or distance code) */
static const uint8_t *text_table();
static inline const uint8_t *rev8()
{
	static uint8_t t[256]; static std::once_flag once;
	std::call_once(once, []() { for (int i = 0; i < 256; ++i) { int r = 0; for (int b = 0; b < 8; ++b) r |= (i >> b & 1) << (7 - b); t[i] = (uint8_t)r; } });
	return t;
}
static int build_table(const uint8_t *lens, int n, int TB, int kind, uint32_t *tab, int cap, bool text_only = false)
{
	const uint8_t *is_text = text_only ? text_table() : 0;
	int count[16] = { 0 }, maxl = 0;
	for (int i = 0; i < n; ++i) { ++count[lens[i]]; if (lens[i] > maxl) maxl = lens[i]; }
	if (maxl == 0) { for (int i = 0; i < (1 << TB); ++i) tab[i] = E_INVALID | 1; return 0; }   /* no code at all: every look-up is invalid (zlib allows the set) */
	int left = 1;
	for (int l = 1; l <= 15; ++l) { left = (left << 1) - count[l]; if (left < 0) return -1; }
	if (left > 0 && (kind == 2 || maxl != 1)) return -1;
	if (left > 0) for (int i = 0; i < (1 << TB); ++i) tab[i] = E_INVALID | 1;   /* (a complete code writes every entry below) */
	uint16_t next[16]; next[0] = 0; next[1] = 0;
	for (int l = 1; l < 15; ++l) next[l + 1] = (uint16_t)((next[l] + count[l]) << 1);
	uint8_t sub_bits[1 << LT_BITS];
	uint16_t code_of[320];
	if (maxl > TB) memset(sub_bits, 0, (size_t)1 << TB);
	for (int s = 0; s < n; ++s) {
		const int l = lens[s];
		if (!l) continue;
		const unsigned c = next[l]++;
		const unsigned r = (unsigned)(rev8()[c & 255] << 8 | rev8()[c >> 8]) >> (16 - l);
		code_of[s] = (uint16_t)r;
		if (l > TB) { uint8_t &sb = sub_bits[r & ((1u << TB) - 1)]; if (l - TB > sb) sb = (uint8_t)(l - TB); }
	}
	int used = 1 << TB;
	if (maxl > TB) for (int i = 0; i < (1 << TB); ++i) if (sub_bits[i]) {
		const int sz = 1 << sub_bits[i];
		if (used + sz > cap) return -1;
		tab[i] = (uint32_t)used << 16 | E_LINK | (uint32_t)sub_bits[i] << 8 | (uint32_t)TB;
		for (int j = 0; j < sz; ++j) tab[used + j] = E_INVALID | 1;
		used += sz;
	}
	for (int s = 0; s < n; ++s) {
		const int l = lens[s];
		if (!l) continue;
		uint32_t e;
		if (kind == 2) e = (uint32_t)s << 16;
		else if (kind == 1) e = s < 30 ? (uint32_t)DIST_BASE[s] << 16 | E_BASE | (uint32_t)DIST_XTRA[s] << 8 : (uint32_t)E_INVALID;
		else if (s < 256) e = is_text && !is_text[s] ? (uint32_t)E_INVALID : (uint32_t)s << 16;   /* (a searched start: a literal that is no text ends the attempt) */
		else if (s == 256) e = E_EOB;
		else e = s < 286 ? (uint32_t)LEN_BASE[s - 257] << 16 | E_BASE | (uint32_t)LEN_XTRA[s - 257] << 8 : (uint32_t)E_INVALID;
		const unsigned r = code_of[s];
		if (l <= TB) { e |= (uint32_t)l; for (unsigned i = r; i < (1u << TB); i += 1u << l) tab[i] = e; }
		else {
			const uint32_t lk = tab[r & ((1u << TB) - 1)];
			const unsigned sb = (lk >> 8) & 15, at = lk >> 16;
			e |= (uint32_t)(l - TB);
			for (unsigned i = r >> TB; i < (1u << sb); i += 1u << (l - TB)) tab[at + i] = e;
		}
	}
	return 0;
}

/* lt: literal / length look-up; an entry of its first 2^LT_BITS that decodes a literal and finds a second whole literal code in the bits left of
 * its index holds both (bit 8 set, the two bytes in the value, the bits of both codes to drop): sequence and quality lines are runs of literals
 * with codes of 2..5 bits, and the look-up -> shift -> look-up chain is what bounds a table-driven decoder.  lt1: the same first level with one
 * symbol per entry, for the last bytes of the input where every symbol must be checked against the bits that are really there */
struct Tables { uint32_t lt[LT_SIZE], dt[DT_SIZE], lt1[1 << LT_BITS]; };
static void pair_literals(Tables &T)
{
	memcpy(T.lt1, T.lt, sizeof(T.lt1));
	for (unsigned i = 0; i < (1u << LT_BITS); ++i) {
		const uint32_t e1 = T.lt1[i];
		if (e1 & 0xf000) continue;
		const unsigned l1 = e1 & 255;
		if (l1 >= LT_BITS) continue;
		const uint32_t e2 = T.lt1[i >> l1];
		if ((e2 & 0xf000) || (e2 & 255) > LT_BITS - l1) continue;
		T.lt[i] = ((e1 >> 16 & 255) | (e2 >> 16 & 255) << 8) << 16 | 0x100 | (l1 + (e2 & 255));
	}
}

static const Tables &fixed_tables()
{
	static Tables *T = 0;
	static std::once_flag once;
	std::call_once(once, []() {
		T = new Tables;
		uint8_t l[288];
		for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
		build_table(l, 288, LT_BITS, 0, T->lt, LT_SIZE);
		for (int i = 0; i < 32; ++i) l[i] = 5;
		build_table(l, 32, DT_BITS, 1, T->dt, DT_SIZE);
		pair_literals(*T);
	});
	return *T;
}

/* the header of a dynamic block behind its three type bits (RFC 1951 3.2.7), with zlib's checks (inflate.c: "too many length or distance
 * symbols", "invalid code lengths set", "invalid bit length repeat", "missing end-of-block", "invalid literal/lengths set", "invalid
 * distances set") */
static int read_dynamic(BitIn &in, Tables &T, bool text_only = false)
{
	static const uint8_t order[19] = { 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 };
	in.refill();
	const int hlit = (int)in.take(5) + 257, hdist = (int)in.take(5) + 1, hclen = (int)in.take(4) + 4;
	if (hlit > 286 || hdist > 30) return D_DATA;
	uint8_t cl[19] = { 0 };
	for (int i = 0; i < hclen; ++i) { if ((i & 7) == 0) in.refill(); cl[order[i]] = (uint8_t)in.take(3); }
	if (in.bc < 0) return D_TRUNC;
	uint32_t ct[1 << 7];
	if (build_table(cl, 19, 7, 2, ct, 1 << 7) != 0) return D_DATA;
	uint8_t lens[320];
	int n = 0;
	while (n < hlit + hdist) {
		in.refill();
		const uint32_t e = ct[in.bb & 127];
		if (e & E_INVALID) return in.bc < 7 && in.p >= in.end ? D_TRUNC : D_DATA;
		in.drop((int)(e & 255));
		const int s = (int)(e >> 16);
		if (s < 16) lens[n++] = (uint8_t)s;
		else {
			int rep, v = 0;
			if (s == 16) { if (n == 0) return D_DATA; v = lens[n - 1]; rep = 3 + (int)in.take(2); }
			else if (s == 17) rep = 3 + (int)in.take(3);
			else rep = 11 + (int)in.take(7);
			if (n + rep > hlit + hdist) return D_DATA;
			while (rep--) lens[n++] = (uint8_t)v;
		}
		if (in.bc < 0) return D_TRUNC;
	}
	if (lens[256] == 0) return D_DATA;
	if (build_table(lens, hlit, LT_BITS, 0, T.lt, LT_SIZE, text_only) != 0) return D_DATA;
	if (build_table(lens + hlit, hdist, DT_BITS, 1, T.dt, DT_SIZE) != 0) return D_DATA;
	pair_literals(T);
	return D_OK;
}

static const uint8_t *text_table()
{
	static uint8_t t[256]; static std::once_flag once;
	std::call_once(once, []() { for (int i = 0; i < 256; ++i) t[i] = (i >= 32 && i <= 126) || i == '\n' || i == '\r' || i == '\t'; });
	return t;
}

/* the symbols of one Huffman-coded block up to its end-of-block.  out[-hist .. at) is what a distance may reach; D_ROOM: `at` came
 * within 272 of `cap`, call again with more room (the stream stands between two symbols).
 * decode_tail: one symbol per step, each checked against the bits the input really holds -- the last 16 bytes of the input.
 * decode_symbols: while more input than that lies ahead a refill always yields 56 true bits: up to three look-ups of literals (each one or
 * two of them) per refill, no checks; a match is copied eight bytes at a time when its distance allows (it may write up to 7 bytes past its
 * end: the room is there) */
template <class T> static inline void copy_match(T *q, size_t dist, unsigned len)
{
	const T *sp = q - dist;
	if (dist * sizeof(T) >= 8) {
		T *const qe = q + len;
		do { memcpy(q, sp, 8); q += 8 / sizeof(T); sp += 8 / sizeof(T); } while (q < qe);
	} else if (dist == 1) {
		const T v = *sp;
		for (unsigned i = 0; i < len; ++i) q[i] = v;
	} else for (unsigned i = 0; i < len; ++i) q[i] = sp[i];
}
template <class T>
static int decode_tail(BitIn &in, const Tables &tb, T *out, size_t &at, size_t cap, size_t hist)
{
	size_t o = at;
	for (;;) {
		if (o + 272 > cap) { at = o; return D_ROOM; }
		in.refill();
		uint32_t e = tb.lt1[in.bb & ((1u << LT_BITS) - 1)];
		if (e & E_LINK) { in.drop(LT_BITS); e = tb.lt[(e >> 16) + (in.bb & ((1u << ((e >> 8) & 15)) - 1))]; }
		in.drop((int)(e & 255));
		if ((e & 0xf000) == 0) {
			if (in.bc < 0) { at = o; return D_TRUNC; }
			out[o++] = (T)(e >> 16);
			continue;
		}
		if (e & E_EOB) { at = o; return in.bc < 0 ? D_TRUNC : D_OK; }
		if (e & (E_INVALID | E_LINK)) { at = o; return in.bc < 0 || (in.p >= in.end && in.bc < 15) ? D_TRUNC : D_DATA; }
		const unsigned lx = (e >> 8) & 15;
		const unsigned len = (e >> 16) + (unsigned)(in.bb & ((1u << lx) - 1));
		in.drop((int)lx);
		uint32_t d = tb.dt[in.bb & ((1u << DT_BITS) - 1)];
		if (d & E_LINK) { in.drop(DT_BITS); d = tb.dt[(d >> 16) + (in.bb & ((1u << ((d >> 8) & 15)) - 1))]; }
		in.drop((int)(d & 255));
		if (d & (E_INVALID | E_LINK)) { at = o; return in.bc < 0 || (in.p >= in.end && in.bc < 15) ? D_TRUNC : D_DATA; }
		const unsigned dx = (d >> 8) & 15;
		const size_t dist = (d >> 16) + (size_t)(in.bb & ((1u << dx) - 1));
		in.drop((int)dx);
		if (in.bc < 0) { at = o; return D_TRUNC; }
		if (dist > o + hist) { at = o; return D_DATA; }           /* zlib: "invalid distance too far back" */
		copy_match(out + o, dist, len);
		o += len;
	}
}
template <class T> static inline void put_literals(T *q, uint32_t e)
{
	if (sizeof(T) == 1) { const uint16_t v = (uint16_t)(e >> 16); memcpy(q, &v, 2); }   /* (the second byte is only kept when the entry holds two) */
	else { q[0] = (T)(e >> 16 & 255); q[1] = (T)(e >> 24); }
}
template <class T>
static int decode_symbols(BitIn &in, const Tables &tb, T *out, size_t &at, size_t cap, size_t hist)
{
	const uint32_t LM = (1u << LT_BITS) - 1;
	size_t o = at;
	for (;;) {
		if (o + 272 > cap) { at = o; return D_ROOM; }
		if (in.end - in.p < 16) { at = o; return decode_tail<T>(in, tb, out, at, cap, hist); }
		in.refill();
		uint32_t e = tb.lt[in.bb & LM];
		if ((e & 0xf000) == 0) {                                 /* literals: the common case in sequence and quality lines */
			in.drop((int)(e & 255)); put_literals(out + o, e); o += 1 + (e >> 8 & 1);
			e = tb.lt[in.bb & LM];
			if ((e & 0xf000) == 0) {
				in.drop((int)(e & 255)); put_literals(out + o, e); o += 1 + (e >> 8 & 1);
				e = tb.lt[in.bb & LM];
				if ((e & 0xf000) == 0) {
					in.drop((int)(e & 255)); put_literals(out + o, e); o += 1 + (e >> 8 & 1);
					continue;
				}
			}
			in.refill();                                          /* what follows may take 48 bits */
			e = tb.lt[in.bb & LM];
		}
		if (e & E_LINK) { in.drop(LT_BITS); e = tb.lt[(e >> 16) + (in.bb & ((1u << ((e >> 8) & 15)) - 1))]; }
		in.drop((int)(e & 255));
		if ((e & 0xf000) == 0) { out[o++] = (T)(e >> 16); continue; }   /* (a literal with a long code) */
		if (e & E_EOB) { at = o; return D_OK; }
		if (e & (E_INVALID | E_LINK)) { at = o; return D_DATA; }
		const unsigned lx = (e >> 8) & 15;
		const unsigned len = (e >> 16) + (unsigned)(in.bb & ((1u << lx) - 1));
		in.drop((int)lx);
		uint32_t d = tb.dt[in.bb & ((1u << DT_BITS) - 1)];
		if (d & E_LINK) { in.drop(DT_BITS); d = tb.dt[(d >> 16) + (in.bb & ((1u << ((d >> 8) & 15)) - 1))]; }
		in.drop((int)(d & 255));
		if (d & (E_INVALID | E_LINK)) { at = o; return D_DATA; }
		const unsigned dx = (d >> 8) & 15;
		const size_t dist = (d >> 16) + (size_t)(in.bb & ((1u << dx) - 1));
		in.drop((int)dx);
		if (dist > o + hist) { at = o; return D_DATA; }           /* zlib: "invalid distance too far back" */
		copy_match(out + o, dist, len);
		o += len;
	}
}

/* CRC-32 of gzip (the reflected polynomial 0xEDB88320) by carry-less multiplication: four 128-bit lanes folded 64 bytes at a time, then folded
 * to one lane, to 64 bits, and reduced (Barrett) -- V. Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction",
 * with the constants of that paper for this polynomial.  zlib 1.2.11's table-driven crc32() does ~1 GB/s, this ~10; the bytes before the first
 * and behind the last whole 16 go through zlib's */
__attribute__((target("pclmul,sse4.1")))
static uint32_t crc32_fold(uint32_t crc, const uint8_t *p, size_t n)   /* n a multiple of 16, >= 64; crc: the running register (not inverted) */
{
	const __m128i R2R1 = _mm_set_epi64x(0x00000001c6e41596ll, 0x0000000154442bd4ll), R4R3 = _mm_set_epi64x(0x00000000ccaa009ell, 0x00000001751997d0ll);
	const __m128i R5 = _mm_set_epi64x(0, 0x0000000163cd6124ll), RU = _mm_set_epi64x(0x00000001F7011641ll, 0x00000001DB710641ll), M32 = _mm_set_epi32(0, 0, 0, -1);
	__m128i x1 = _mm_loadu_si128((const __m128i*)p), x2 = _mm_loadu_si128((const __m128i*)(p + 16)), x3 = _mm_loadu_si128((const __m128i*)(p + 32)), x4 = _mm_loadu_si128((const __m128i*)(p + 48));
	x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
	p += 64; n -= 64;
	while (n >= 64) {
		__m128i t1 = _mm_clmulepi64_si128(x1, R2R1, 0x11), t2 = _mm_clmulepi64_si128(x2, R2R1, 0x11), t3 = _mm_clmulepi64_si128(x3, R2R1, 0x11), t4 = _mm_clmulepi64_si128(x4, R2R1, 0x11);
		x1 = _mm_clmulepi64_si128(x1, R2R1, 0x00); x2 = _mm_clmulepi64_si128(x2, R2R1, 0x00); x3 = _mm_clmulepi64_si128(x3, R2R1, 0x00); x4 = _mm_clmulepi64_si128(x4, R2R1, 0x00);
		x1 = _mm_xor_si128(_mm_xor_si128(x1, t1), _mm_loadu_si128((const __m128i*)p));
		x2 = _mm_xor_si128(_mm_xor_si128(x2, t2), _mm_loadu_si128((const __m128i*)(p + 16)));
		x3 = _mm_xor_si128(_mm_xor_si128(x3, t3), _mm_loadu_si128((const __m128i*)(p + 32)));
		x4 = _mm_xor_si128(_mm_xor_si128(x4, t4), _mm_loadu_si128((const __m128i*)(p + 48)));
		p += 64; n -= 64;
	}
	__m128i t;
	t = _mm_clmulepi64_si128(x1, R4R3, 0x11); x1 = _mm_clmulepi64_si128(x1, R4R3, 0x00); x1 = _mm_xor_si128(_mm_xor_si128(x1, t), x2);
	t = _mm_clmulepi64_si128(x1, R4R3, 0x11); x1 = _mm_clmulepi64_si128(x1, R4R3, 0x00); x1 = _mm_xor_si128(_mm_xor_si128(x1, t), x3);
	t = _mm_clmulepi64_si128(x1, R4R3, 0x11); x1 = _mm_clmulepi64_si128(x1, R4R3, 0x00); x1 = _mm_xor_si128(_mm_xor_si128(x1, t), x4);
	while (n >= 16) {
		t = _mm_clmulepi64_si128(x1, R4R3, 0x11); x1 = _mm_clmulepi64_si128(x1, R4R3, 0x00);
		x1 = _mm_xor_si128(_mm_xor_si128(x1, t), _mm_loadu_si128((const __m128i*)p));
		p += 16; n -= 16;
	}
	t = _mm_clmulepi64_si128(R4R3, x1, 0x01);                      /* 128 -> 64 bits: R4 x the low half */
	x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
	__m128i x2b = _mm_srli_si128(x1, 4);                            /* 64 -> 32 */
	x1 = _mm_and_si128(x1, M32);
	x1 = _mm_clmulepi64_si128(x1, R5, 0x00);
	x1 = _mm_xor_si128(x1, x2b);
	x2b = x1;                                                       /* Barrett */
	x1 = _mm_and_si128(x1, M32);
	x1 = _mm_clmulepi64_si128(x1, RU, 0x10);
	x1 = _mm_and_si128(x1, M32);
	x1 = _mm_clmulepi64_si128(x1, RU, 0x00);
	x1 = _mm_xor_si128(x1, x2b);
	return (uint32_t)_mm_extract_epi32(x1, 1);
}
static uint32_t crc32_bytes(uint32_t crc, const uint8_t *p, size_t n)   /* zlib's crc32(crc, p, n) */
{
	static const bool fast = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !tune().no_simd;
	if (fast && n >= 256) {
		const size_t body = n & ~(size_t)15;
		crc = ~crc32_fold(~crc, p, body);
		p += body; n -= body;
	}
	while (n) { const size_t step = std::min<size_t>(n, (size_t)1 << 30); crc = (uint32_t)crc32(crc, p, (uInt)step); p += step; n -= step; }
	return crc;
}

/* one member's end or the stream's: what follows a final block */
struct Mark { size_t out_at; uint32_t crc, isize; };

/* output of a decode: bytes behind a known window, or symbols behind an unknown one.  v[0 .. WSIZE) is the window */
template <class T> struct Out {
	std::vector<T> v; size_t n, hist;
	Out() : n(0), hist(0) {}
	T *data() { return v.data() + WSIZE; }
	void room(size_t want) { if (v.size() < WSIZE + want) v.resize(WSIZE + want + (want >> 1)); }
};

struct Run {                                                   /* what one decode run reports */
	uint64_t start, end;                                       /* first bit, bit behind the last whole block (and trailer / next header) */
	int rc;                                                    /* D_OK; D_TRUNC (the input ended: `end` is where); D_DATA (at `end`) */
	bool stream_end;                                           /* the last member ended, or nothing decodable follows */
	std::vector<Mark> marks;
	Run() : start(0), end(0), rc(D_OK), stream_end(false) {}
};

/* behind a final block: the member's trailer, then the next member's header, zlib's gz_look()/gz_head() way -- anything that is not a gzip header
 * ends the stream ("trailing garbage is ignored").  `bit` is the bit behind the final block; returns the bit of the next member's first block */
static bool next_member(const uint8_t *in, size_t n, uint64_t bit, uint32_t *crc, uint32_t *isize, uint64_t *next_bit, bool *trunc, bool *bad)
{
	size_t p = (size_t)((bit + 7) >> 3);
	*trunc = *bad = false;
	if (p + 8 > n) { *trunc = true; return false; }
	*crc = in[p] | in[p + 1] << 8 | in[p + 2] << 16 | (uint32_t)in[p + 3] << 24;
	*isize = in[p + 4] | in[p + 5] << 8 | in[p + 6] << 16 | (uint32_t)in[p + 7] << 24;
	p += 8;
	*next_bit = (uint64_t)p * 8;
	if (p + 2 > n || in[p] != 0x1f || in[p + 1] != 0x8b) return false;
	if (p + 4 <= n && (in[p + 2] != 8 || (in[p + 3] & 0xe0))) { *bad = true; return false; }   /* the magic commits zlib to a member: "unknown compression method" / "unknown header flags set" */
	if (p + 10 > n) return false;
	const int flg = in[p + 3];
	p += 10;
	if (flg & 4) { if (p + 2 > n) return false; p += 2 + (in[p] | in[p + 1] << 8); }
	if (flg & 8) { while (p < n && in[p]) ++p; ++p; }
	if (flg & 16) { while (p < n && in[p]) ++p; ++p; }
	if (flg & 2) p += 2;
	if (p >= n) return false;
	*next_bit = (uint64_t)p * 8;
	return true;
}

/* the first member's header; the bit its first block starts at, or 0 if this is not gzip */
static uint64_t first_member(const uint8_t *in, size_t n)
{
	if (n < 18 || in[0] != 0x1f || in[1] != 0x8b || in[2] != 8 || (in[3] & 0xe0)) return 0;
	const int flg = in[3];
	size_t p = 10;
	if (flg & 4) { if (p + 2 > n) return 0; p += 2 + (in[p] | in[p + 1] << 8); }
	if (flg & 8) { while (p < n && in[p]) ++p; ++p; }
	if (flg & 16) { while (p < n && in[p]) ++p; ++p; }
	if (flg & 2) p += 2;
	return p < n ? (uint64_t)p * 8 : 0;
}

/* whole blocks from `bit` on, until one would start at or behind `limit` (the first one is always decoded), the stream ends, or an error.
 * TEXT: only for the blocks of a searched start (a literal outside text ends the run as a data error would) */
template <class T, bool TEXT>
static void decode_run(const uint8_t *inp, size_t n, uint64_t bit, uint64_t limit, Out<T> &O, Tables &T_, Run &R)
{
	BitIn in;
	R.start = R.end = bit; R.rc = D_OK; R.stream_end = false; R.marks.clear();
	bool first = true;
	for (;;) {
		if (!first && bit >= limit) break;
		first = false;
		if ((bit >> 3) >= n) { R.rc = D_TRUNC; break; }
		in.init(inp, n, bit);
		const size_t n0 = O.n;
		const int bfinal = (int)in.take(1), btype = (int)in.take(2);
		int rc = in.bc < 0 ? D_TRUNC : D_OK;
		if (rc == D_OK && btype == 3) rc = D_DATA;
		if (rc == D_OK && btype == 0) {
			size_t p = (size_t)((in.bit_pos() + 7) >> 3);
			if (p + 4 > n) rc = D_TRUNC;
			else {
				const unsigned len = inp[p] | inp[p + 1] << 8, nlen = inp[p + 2] | inp[p + 3] << 8;
				if ((len ^ 0xffff) != nlen) rc = D_DATA;
				else {
					p += 4;
					const size_t take = std::min<size_t>(len, n - p);
					O.room(O.n + take + 272);
					T *o = O.data() + O.n;
					for (size_t i = 0; i < take; ++i) o[i] = (T)inp[p + i];
					O.n += take;
					if (take < len) rc = D_TRUNC;                       /* zlib delivers the bytes that are there */
					else in.init(inp, n, (uint64_t)(p + len) * 8);
				}
			}
		} else if (rc == D_OK) {
			const Tables *tb = &fixed_tables();
			if (btype == 2) { rc = read_dynamic(in, T_, TEXT); tb = &T_; }
			while (rc == D_OK) {
				O.room(O.n + (1 << 16));
				rc = decode_symbols<T>(in, *tb, O.data(), O.n, O.v.size() - WSIZE, O.hist);
				if (rc == D_ROOM) { rc = D_OK; continue; }
				break;
			}
		}
		if (rc != D_OK && TEXT) { O.n = n0; R.end = bit; return; }          /* a run of symbols just ends before the block: the stitch decodes on from there, exactly, and meets what there is to meet */
		if (rc == D_TRUNC) { R.rc = rc; R.end = bit; return; }               /* the symbols before the cut stay (gzread delivers them); nothing follows */
		if (rc != D_OK) { O.n = n0; R.rc = rc; R.end = bit; return; }
		bit = in.bit_pos();
		R.end = bit;
		if (bfinal) {
			Mark m; m.out_at = O.n; m.crc = m.isize = 0;
			bool trunc = false, bad = false;
			uint64_t nb = bit;
			const bool more = next_member(inp, n, bit, &m.crc, &m.isize, &nb, &trunc, &bad);
			if (trunc) { R.rc = D_TRUNC; R.stream_end = true; return; }   /* the trailer is cut: gzread delivers the data and reports the end */
			R.marks.push_back(m);
			R.end = bit = nb;
			if (bad) { R.rc = D_DATA; R.stream_end = true; return; }      /* 1f 8b followed by a method / flags zlib refuses */
			O.hist = (size_t)0 - O.n;                             /* a new member starts with an empty window: o + hist = the bytes of THIS member in front of o (modulo 2^64) ... */
			if (!more) { R.stream_end = true; return; }
			if (sizeof(T) == 2) return;                             /* ... which a run of symbols cannot express: the stitch goes on from here, exactly */
		}
	}
}

/* the first bit in [from, to) where a block that passes for the start of a text stream's dynamic block begins; ~0 if none */
struct Searcher {
	Tables T, T2;
	Out<uint16_t> probe;
	uint64_t find(const uint8_t *in, size_t n, uint64_t from, uint64_t to) {
		if (n < 64) return ~0ull;
		to = std::min<uint64_t>(to, (uint64_t)(n - 32) * 8);
		probe.room(1 << 17);
		for (uint64_t p = from; p < to; ++p) {
			const uint8_t *q = in + (p >> 3);
			const uint64_t v = ld64(q) >> (p & 7);
			if ((v & 7) != 4) continue;                            /* BFINAL = 0, BTYPE = 2 */
			if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) continue;
			const int hclen = (int)((v >> 13) & 15) + 4;
			const uint64_t w = ld64(q + 2) >> ((p & 7) + 1);       /* the code-length code lengths start 17 bits in */
			const uint64_t w2 = ld64(q + 8) >> ((p & 7) + 1);     /* ... and run for up to 57 bits: bits 48.. of them */
			int left = 128;
			for (int i = 0; i < hclen; ++i) {
				const int b = 3 * i;
				const int l = (int)((b < 48 ? w >> b : w2 >> (b - 48)) & 7);
				if (l) left -= 128 >> l;
			}
			if (left != 0) continue;                               /* the code-length code must be complete */
			BitIn bi; bi.init(in, n, p + 3);
			if (read_dynamic(bi, T, true) != D_OK) continue;
			/* the block itself: text only, every distance inside the (unknown) window */
			probe.n = 0; probe.hist = WSIZE;
			int rc;
			for (;;) {
				probe.room(probe.n + (1 << 16));
				rc = decode_symbols<uint16_t>(bi, T, probe.data(), probe.n, probe.v.size() - WSIZE, WSIZE);
				if (rc != D_ROOM) break;
			}
			if (rc != D_OK || probe.n < 32) continue;
			/* what follows must look like a block too */
			bi.refill();
			const int bt = (int)((bi.bb >> 1) & 3);
			if (bi.bc < 3 || bt == 3) continue;
			if (bt == 2) { BitIn b2 = bi; b2.drop(3); if (read_dynamic(b2, T2) != D_OK) continue; }
			else if (bt == 0) {
				const size_t s = (size_t)((bi.bit_pos() + 3 + 7) >> 3);
				if (s + 4 > n || ((in[s] | in[s + 1] << 8) ^ 0xffff) != (in[s + 2] | in[s + 3] << 8)) continue;
			}
			return p;
		}
		return ~0ull;
	}
};

struct Worker {
	Searcher se;
	Tables T;
	Out<uint16_t> sym;
	Out<uint8_t> bytes;
	Run run;
	bool found;
	Worker() : found(false) {}
};

/* a piece of a batch's output, in stream order */
struct Piece {
	int kind;                                                  /* 0: bytes of worker `src`; 1: symbols of worker `src`; 2: bytes of gap buffer `src` */
	int src;
	size_t len, dst;                                           /* bytes, and where they go in the batch */
	std::vector<uint8_t> win;                                  /* kind 1: the WSIZE bytes before it (position WSIZE - 1 is the last one) */
	std::vector<Mark> marks;
	std::vector<uint32_t> crc;                                 /* of the stretches the marks cut the piece into (marks.size() + 1 of them) */
};


struct Reader {
	int fd; const uint8_t *in; size_t n;
	int n_thr;
	uint64_t pos;                                              /* the bit the next batch starts at */
	std::vector<uint8_t> win;                                  /* the (up to) WSIZE bytes before it */
	bool stream_end, failed; std::string why;
	uint32_t m_crc; uint64_t m_len;                            /* CRC32 and length of the current member so far */
	std::vector<Worker*> wk;
	std::vector<Out<uint8_t>*> gaps;
	/* two batch buffers, [room for what the consumer carries over][the batch]: while the consumer works on one the producer fills the other */
	struct Buf { uint8_t *p; size_t cap, off, len; bool last; Buf() : p(0), cap(0), off(0), len(0), last(false) {} } buf[2];
	size_t R;
	std::thread producer;
	int k_made, k_taken;                                       /* batches produced (or being produced) / handed out */
	uint8_t *cur_ptr; size_t cur_len;                          /* the batch handed out last, carry included */
	uint64_t n_search_ok, n_search_bad, n_gap_bits;             /* statistics: chunks accepted, chunks decoded again, bits the stitch decoded itself */

	Reader() : fd(-1), in(0), n(0), n_thr(1), pos(0), stream_end(false), failed(false), m_crc(0), m_len(0), R(tune().front), k_made(0), k_taken(0),
	           cur_ptr(0), cur_len(0), n_search_ok(0), n_search_bad(0), n_gap_bits(0) {}
	Reader(const Reader&) = delete; Reader &operator=(const Reader&) = delete;
	~Reader() { close(); }

	/* false: not a file this reader takes (not regular, not gzip, too small) -- the caller keeps its gzread() path */
	bool open(const char *fn, int threads, bool any_size = false) {
		struct stat sb;
		if (stat(fn, &sb) != 0 || !S_ISREG(sb.st_mode)) return false;   /* (before open(): opening a FIFO whose writer has gone blocks for ever) */
		fd = ::open(fn, O_RDONLY);
		if (fd < 0) return false;
		if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 18 || (!any_size && (size_t)sb.st_size < tune().min_size)) { ::close(fd); fd = -1; return false; }
		void *m = mmap(0, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
		if (m == MAP_FAILED) { ::close(fd); fd = -1; return false; }
		in = (const uint8_t*)m; n = (size_t)sb.st_size;
		(void)madvise(m, n, MADV_SEQUENTIAL);
		pos = first_member(in, n);
		if (pos == 0) { close(); return false; }
		n_thr = std::max(1, std::min(threads, 64));
		for (int i = 0; i < n_thr; ++i) wk.push_back(new Worker);
		m_crc = (uint32_t)crc32(0L, Z_NULL, 0); m_len = 0;
		producer = std::thread([this]() { produce(0); });
		k_made = 1;
		return true;
	}
	void close() {
		if (producer.joinable()) producer.join();
		for (auto *w : wk) delete w;
		wk.clear();
		for (auto *g : gaps) delete g;
		gaps.clear();
		for (int i = 0; i < 2; ++i) { free(buf[i].p); buf[i] = Buf(); }
		if (in) munmap((void*)in, n);
		if (fd >= 0) ::close(fd);
		in = 0; fd = -1;
	}

	/* the next batch, with the bytes from `keep_from` on of the one before it in front: *ptr .. *ptr + *len; *last: the stream ends with it.
	 * false: the reader failed (why) */
	bool next(size_t keep_from, uint8_t **ptr, size_t *len, bool *last) {
		const uint8_t *carry = 0; size_t carry_len = 0;
		if (k_taken > 0) { keep_from = std::min(keep_from, cur_len); carry = cur_ptr + keep_from; carry_len = cur_len - keep_from; }
		if (producer.joinable()) producer.join();
		if (failed) return false;
		Buf &b = buf[k_taken & 1];
		if (carry_len > b.off) {                                   /* (a sequence longer than the room in front: the batch moves) */
			const size_t off = carry_len + (carry_len >> 2) + ((size_t)1 << 20);
			uint8_t *np = (uint8_t*)malloc(off + b.len + 64);
			if (!np) { fail("out of memory"); return false; }
			memcpy(np + off, b.p + b.off, b.len);
			free(b.p); b.p = np; b.off = off; b.cap = off + b.len + 64;
		}
		if (carry_len) memcpy(b.p + b.off - carry_len, carry, carry_len);
		cur_ptr = b.p + b.off - carry_len; cur_len = carry_len + b.len;
		*ptr = cur_ptr; *len = cur_len; *last = b.last;
		++k_taken;
		if (!b.last) { const int k = k_made++; producer = std::thread([this, k]() { produce(k); }); }   /* into the buffer the carry was just copied out of */
		return true;
	}

	void fail(const std::string &s) { failed = true; why = s; }

	/* CRC32 / ISIZE of the member that ends at a mark */
	bool member_ends(const Mark &m) {
		if (m.crc != m_crc || m.isize != (uint32_t)m_len) { fail(m.crc != m_crc ? "gzip: incorrect data check (CRC32)" : "gzip: incorrect length check (ISIZE)"); return false; }
		m_crc = (uint32_t)crc32(0L, Z_NULL, 0); m_len = 0;
		return true;
	}

	void produce(int k) {
		Buf &b = buf[k & 1];
		b.len = 0; b.last = false;
		if (b.off != R) { free(b.p); b.p = 0; b.cap = 0; b.off = R; }         /* (a buffer that was moved for a long carry goes back to the usual layout) */
		const size_t CH = std::max<size_t>(tune().chunk, 1024);
		const size_t b0 = (size_t)(pos >> 3);
		const int Tn = (int)std::min<size_t>((size_t)n_thr, std::max<size_t>(1, (n - b0 + CH - 1) / CH));
		const uint64_t batch_end = (uint64_t)std::min(n, b0 + (size_t)Tn * CH) * 8;
		auto cut = [&](int c) { return c >= Tn ? batch_end : (uint64_t)(b0 + (size_t)c * CH) * 8; };
		/* 1. every chunk on its own thread */
		{
			Worker &w0 = *wk[0];
			w0.bytes.room(WSIZE);
			memset(w0.bytes.v.data(), 0, WSIZE);
			memcpy(w0.bytes.v.data() + WSIZE - win.size(), win.data(), win.size());
			w0.bytes.n = 0; w0.bytes.hist = win.size();
		}
		std::vector<std::thread> th;
		auto work = [&](int c) {
			Worker &w = *wk[c];
			if (c == 0) { decode_run<uint8_t, false>(in, n, pos, cut(1), w.bytes, w.T, w.run); w.found = true; return; }
			w.found = false;
			const uint64_t s = w.se.find(in, n, cut(c), cut(c + 1));
			if (s == ~0ull) return;
			w.sym.room(CH * 4);
			uint16_t *v = w.sym.v.data();
			for (int i = 0; i < WSIZE; ++i) v[i] = (uint16_t)(256 + i);
			w.sym.n = 0; w.sym.hist = WSIZE;
			decode_run<uint16_t, true>(in, n, s, cut(c + 1), w.sym, w.T, w.run);
			w.found = w.run.end > w.run.start;
		};
		for (int c = 1; c < Tn; ++c) th.emplace_back(work, c);
		work(0);
		for (auto &t : th) t.join();
		th.clear();
		/* 2. the stitch */
		std::vector<Piece> pieces;
		size_t n_gap = 0;
		bool ended = false;
		auto push_window = [&](const uint8_t *p, size_t len) {    /* win = the last WSIZE bytes of win + p[0 .. len) */
			if (len >= WSIZE) { win.assign(p + len - WSIZE, p + len); return; }
			if (win.size() + len > WSIZE) win.erase(win.begin(), win.begin() + (win.size() + len - WSIZE));
			win.insert(win.end(), p, p + len);
		};
		auto take_bytes = [&](int kind, int src, Out<uint8_t> &O, Run &r) -> bool {
			Piece pc; pc.kind = kind; pc.src = src; pc.len = O.n; pc.dst = 0; pc.marks = r.marks;
			size_t at = 0;
			for (const Mark &m : r.marks) { push_window(O.data() + at, m.out_at - at); win.clear(); at = m.out_at; }
			push_window(O.data() + at, O.n - at);
			pieces.push_back(std::move(pc));
			pos = r.end;
			if (r.rc == D_DATA) { fail("gzip: invalid deflate data"); return false; }
			if (r.rc == D_TRUNC || r.stream_end) ended = true;
			return true;
		};
		if (!take_bytes(0, 0, wk[0]->bytes, wk[0]->run)) return;
		auto fill_gap = [&](uint64_t want) -> bool {                 /* the bits nobody decoded, or somebody decoded from a false start */
			while (pos < want && !ended) {
				if (n_gap == gaps.size()) gaps.push_back(new Out<uint8_t>);
				Out<uint8_t> &G = *gaps[n_gap];
				G.room(WSIZE);
				memset(G.v.data(), 0, WSIZE);
				memcpy(G.v.data() + WSIZE - win.size(), win.data(), win.size());
				G.n = 0; G.hist = win.size();
				Run r;
				const uint64_t p0 = pos;
				decode_run<uint8_t, false>(in, n, pos, want, G, wk[0]->T, r);
				if (!take_bytes(2, (int)n_gap, G, r)) return false;
				++n_gap;
				n_gap_bits += pos - p0;
			}
			return true;
		};
		for (int c = 1; c < Tn && !ended; ++c) {
			Worker &w = *wk[c];
			if (!w.found || w.run.start < pos) { ++n_search_bad; continue; }   /* nothing usable: a later gap covers its range */
			if (!fill_gap(w.run.start)) return;
			if (ended) break;
			if (pos != w.run.start) { ++n_search_bad; continue; }    /* the exact decode ran past its start: a false one */
			++n_search_ok;
			Piece pc; pc.kind = 1; pc.src = c; pc.len = w.sym.n; pc.dst = 0; pc.marks = w.run.marks;
			pc.win.assign(WSIZE, 0);
			memcpy(pc.win.data() + WSIZE - win.size(), win.data(), win.size());
			{                                                       /* its last WSIZE bytes, resolved, are the window behind it */
				const uint16_t *s = w.sym.data();
				const size_t len = w.sym.n, from = len > WSIZE ? len - WSIZE : 0;
				std::vector<uint8_t> tail(len - from);
				for (size_t i = from; i < len; ++i) { const uint16_t x = s[i]; tail[i - from] = x < 256 ? (uint8_t)x : pc.win[x - 256]; }
				if (!w.run.marks.empty()) { win.clear(); tail.clear(); }   /* (a run of symbols ends with the member: the next one starts with no window) */
				push_window(tail.data(), tail.size());
			}
			pieces.push_back(std::move(pc));
			pos = w.run.end;
			if (w.run.rc == D_DATA) { fail("gzip: invalid deflate data"); return; }   /* (only a member header zlib refuses: any other error just ends a run of symbols) */
			if (w.run.stream_end) ended = true;
		}
		if (!ended && !fill_gap(batch_end)) return;
		/* 3. translate + CRC32, every piece on a thread of its own */
		size_t total = 0;
		for (Piece &pc : pieces) { pc.dst = total; total += pc.len; }
		if (b.cap < R + total + 64) { free(b.p); b.cap = R + total + (total >> 3) + 64; b.p = (uint8_t*)malloc(b.cap); if (!b.p) { b.cap = 0; fail("out of memory"); return; } }
		uint8_t *dst = b.p + R;
		auto emit = [&](size_t i) {
			Piece &pc = pieces[i];
			uint8_t *o = dst + pc.dst;
			if (pc.kind == 1) {
				const uint16_t *s = wk[pc.src]->sym.data();
				std::vector<uint8_t> lut(256 + WSIZE);
				for (int j = 0; j < 256; ++j) lut[j] = (uint8_t)j;
				memcpy(lut.data() + 256, pc.win.data(), WSIZE);
				size_t j = 0;
				const __m128i zero = _mm_setzero_si128();
				for (; j + 16 <= pc.len; j += 16) {                 /* sixteen symbols that are all literals are just narrowed */
					const __m128i a = _mm_loadu_si128((const __m128i*)(s + j)), b2 = _mm_loadu_si128((const __m128i*)(s + j + 8));
					if (_mm_movemask_epi8(_mm_cmpeq_epi16(_mm_srli_epi16(_mm_or_si128(a, b2), 8), zero)) == 0xFFFF) _mm_storeu_si128((__m128i*)(o + j), _mm_packus_epi16(a, b2));
					else for (int q = 0; q < 16; ++q) o[j + q] = lut[s[j + q]];
				}
				for (; j < pc.len; ++j) o[j] = lut[s[j]];
			} else memcpy(o, pc.kind == 0 ? wk[pc.src]->bytes.data() : gaps[pc.src]->data(), pc.len);
			size_t at = 0;
			for (size_t m = 0; m <= pc.marks.size(); ++m) {
				const size_t e = m < pc.marks.size() ? pc.marks[m].out_at : pc.len;
				pc.crc.push_back(crc32_bytes((uint32_t)crc32(0L, Z_NULL, 0), o + at, e - at));
				at = e;
			}
		};
		{
			size_t next_i = 0; std::mutex mu;
			auto loop = [&]() { for (;;) { size_t i; { std::lock_guard<std::mutex> g(mu); if (next_i >= pieces.size()) return; i = next_i++; } emit(i); } };
			const int nt = (int)std::min<size_t>((size_t)n_thr, pieces.size());
			for (int t = 1; t < nt; ++t) th.emplace_back(loop);
			loop();
			for (auto &t : th) t.join();
		}
		for (Piece &pc : pieces) {
			size_t at = 0;
			for (size_t m = 0; m <= pc.marks.size(); ++m) {
				const size_t e = m < pc.marks.size() ? pc.marks[m].out_at : pc.len;
				if (e > at) { m_crc = (uint32_t)crc32_combine(m_crc, pc.crc[m], (z_off_t)(e - at)); m_len += e - at; }
				if (m < pc.marks.size() && !member_ends(pc.marks[m])) return;
				at = e;
			}
		}
		b.len = total;
		b.last = ended || (pos >> 3) >= n;
		stream_end = b.last;
		if (b.last) {                                              /* the decoders' buffers go back now, while the consumer still parses and counts */
			for (auto *w : wk) delete w;
			wk.clear();
			for (auto *g : gaps) delete g;
			gaps.clear();
		}
	}
};

} /* namespace pgz */
#endif
