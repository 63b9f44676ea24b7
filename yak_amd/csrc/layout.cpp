/*
 * layout.cpp -- the exact-layout stage of a pass: capacities after the new keys (khashl.h:197-221), the doubling / placement schedule of the
 * streaming replay (kern_replay2.inc) and the one-workgroup-per-table replay of small tables (k_replay).  Cut out of engine.cpp in round 6.
 */
#include "engine_int.h"

/* ------------------------------------------------------------------------------------------
 * layout planning + replay
 * ------------------------------------------------------------------------------------------ */

/* capacity after `m` new keys on a table of (cap, cnt), plus one possible trailing doubling */
static u32 plan_cap(u32 cap, u32 cnt, u32 m, bool may_trail)
{
	u64 n = cap, c = cnt, rem = m;
	while (rem > 0) {
		const u64 thr = (n >> 1) + (n >> 2);
		if (c >= thr) { n = n ? n << 1 : 4; continue; }
		const u64 b = std::min(rem, thr - c);
		c += b; rem -= b;
	}
	if (may_trail && c >= (n >> 1) + (n >> 2)) n = n ? n << 1 : 4;
	return (u32)n;
}

/* rebuild the image from per-sub-table ordered record lists.  rec_t/lastput may be NULL (shrink) */
/* launch parameters of k_replay for a set of tasks (shared by the two replay drivers) */
static void legacy_replay_launch(yakamd_ctx *c, const std::vector<ReplayTask> &tasks, const ReplayTask *d_tasks, u64 *nk, u32 *nu, u32 *su, u32 *so, u64 *sp,
                                 const u64 *d_rec_kc, const u64 *d_rec_t, const u64 *d_lastput, u32 *d_ob, u32 *d_oc)
{
	const int P = c->P, n_active = c->phi - c->plo;
	u32 cap_top = 0;
	for (int p = 0; p < P; ++p) if (tasks[p].m) cap_top = std::max(cap_top, 1u << tasks[p].cap_max_bits);
	u32 lds_words = std::min<u32>(cap_top, (u32)env_i64("YAKAMD_REPLAY_LDS", 16384));   /* 64 KB: two workgroups per CU (measured 18.5 ms against 20.5 with 128 KB); 0: owner ranks in global scratch */
	int n_thr = n_active <= 256 ? 1024 : n_active <= 512 ? 512 : 256;
	if (lds_words * 4 >= 96 * 1024) n_thr = 1024; else if (lds_words * 4 >= 48 * 1024) n_thr = std::max(n_thr, 512);
	n_thr = (int)env_i64("YAKAMD_REPLAY_THREADS", n_thr);
	yk_launch_replay(d_tasks, P, n_thr, c->d_keys, c->d_used, nk, nu, su, so, sp, d_rec_kc, d_rec_t, d_lastput, d_ob, d_oc, lds_words, c->st);
}

/* Layout replay with the large sub-tables on the streaming kernels (kernels.hip "replay2").  A sub-table whose
 * final capacity stays within 2^SB slots is replayed by k_replay as before.  A larger one is brought to 2^SB slots
 * by k_replay (everything in LDS there), then all of them advance together, step by step: a placement of the next
 * keys up to the growth threshold, or a doubling.  The schedule is khashl's (khashl.h:202: grow BEFORE the put once
 * count >= 0.75 capacity; a trailing put-call on an existing key can still double) and is simulated here on the
 * host; the kernels only move keys.  Returns 0 done, -1 error, 1 not applicable / refused (caller: k_replay). */
static u32 g_r2_used = 0, g_r2_refused = 0;        /* debug counters: replays done by the streaming kernels / handed back to k_replay */

static int run_replay_v2(yakamd_ctx *c, const std::vector<u32> &m, const u64 *d_rec_kc, const u64 *d_rec_t,
                         const u64 *d_lastput, const std::vector<u32> *init_bits, bool from_empty, const std::vector<u64> *rec_off_in)
{
	const int P = c->P;
	if (env_i64("YAKAMD_REPLAY2", 1) == 0 || P > 65536) return 1;
	const u32 SB = (u32)std::min<int64_t>(20, std::max<int64_t>(5, env_i64("YAKAMD_R2_SMALL_BITS", 13))), SMALLCAP = 1u << SB;
	std::vector<ReplayTask> tasks(P);
	std::vector<u64> rec_off(P), new_off(P);
	std::vector<u32> cap0(P), cnt0(P), capm(P);
	std::vector<char> large(P, 0);
	u64 rec = 0, tot = 0;
	bool any = false;
	for (int p = 0; p < P; ++p) {
		const u32 ob = from_empty ? YK_NOCAP : c->h_bits[p], ib = init_bits ? (*init_bits)[p] : YK_NOCAP;
		cnt0[p] = from_empty ? 0 : c->h_count[p];
		cap0[p] = ob == YK_NOCAP ? 0 : 1u << ob;
		if (cap0[p] == 0 && ib != YK_NOCAP) cap0[p] = 1u << ib;
		rec_off[p] = rec_off_in ? (*rec_off_in)[p] : rec; rec += m[p];
		capm[p] = plan_cap(cap0[p], cnt0[p], m[p], d_lastput != 0);
		large[p] = capm[p] > SMALLCAP;
		any = any || large[p];
		new_off[p] = tot; tot += std::max<u64>(32, capm[p]);
	}
	if (!any) return 1;
	/* trailing put-calls (device data: last put-call and the time of the last new key per sub-table) */
	std::vector<u32> trail(P, 0);
	std::vector<u64> lp_host(P, 0);
	u32 *d_m = 0, *d_trail = 0; u64 *d_ro = 0, *d_lp2 = 0;
	u64 *nk = 0, *sp = 0, *K0 = 0, *K1 = 0, *pk = 0, *spill = 0; u32 *nu = 0, *su = 0, *so = 0, *d_ob = 0, *d_oc = 0, *TAG = 0, *OCC = 0, *USED = 0, *pcnt = 0, *pr = 0, *segst = 0, *head = 0, *Fc = 0, *misc = 0;
	ReplayTask *d_tasks = 0; R2Tab *d_tabs = 0; R2Act *d_acts = 0; R2Load *d_ld = 0; R2Pub *d_pub = 0;
	struct Guard { std::function<void()> f; ~Guard() { f(); } } guard{ [&]() {
		dfree(d_m); dfree(d_trail); dfree(d_ro); dfree(d_lp2); dfree(nk); dfree(sp); dfree(K0); dfree(K1); dfree(pk); dfree(spill); dfree(nu); dfree(su); dfree(so);
		dfree(d_ob); dfree(d_oc); dfree(TAG); dfree(OCC); dfree(USED); dfree(pcnt); dfree(pr); dfree(segst); dfree(head); dfree(Fc); dfree(misc); dfree(d_tasks); dfree(d_tabs); dfree(d_acts); dfree(d_ld); dfree(d_pub);
	} };
	if (d_lastput) {
		if (dmalloc(&d_m, P) || dmalloc(&d_trail, P) || dmalloc(&d_ro, P) || dmalloc(&d_lp2, P)) return -1;
		HIPCK(hipMemcpyAsync(d_m, m.data(), P * 4, hipMemcpyHostToDevice, c->st));
		HIPCK(hipMemcpyAsync(d_ro, rec_off.data(), P * 8, hipMemcpyHostToDevice, c->st));
		yk_r2_trail(d_lastput, d_rec_t, d_ro, d_m, P, d_trail, c->st);
		HIPCK(hipMemcpyAsync(trail.data(), d_trail, P * 4, hipMemcpyDeviceToHost, c->st));
		HIPCK(hipMemcpyAsync(lp_host.data(), d_lastput, P * 8, hipMemcpyDeviceToHost, c->st));
		HIPCK(hipStreamSynchronize(c->st));
	}
	/* the schedule of every large sub-table: keys placed by k_replay first (m1), then its actions */
	struct Act { u32 kind, bits, i0, batch; };
	std::vector<std::vector<Act> > sched(P);
	std::vector<u32> m1(P, 0), bitsS(P, YK_NOCAP), cntS(P, 0), bitsF(P, YK_NOCAP), cntF(P, 0);
	size_t n_steps = 0;
	u32 n_large = 0;
	for (int p = 0; p < P; ++p) {
		if (!large[p]) continue;
		++n_large;
		u64 cap = cap0[p], cnt = cnt0[p], rem = m[p];
		while (rem > 0 && cap0[p] <= SMALLCAP) {                   /* the part k_replay does: up to a full table of SMALLCAP slots */
			const u64 thr = (cap >> 1) + (cap >> 2);
			if (cnt >= thr) { if (cap >= SMALLCAP) break; cap = cap ? cap << 1 : 4; continue; }
			const u64 b = std::min(rem, thr - cnt);
			cnt += b; rem -= b;
		}
		m1[p] = (u32)(m[p] - rem);
		bitsS[p] = cap ? (u32)ceil_log2_u64(cap) : YK_NOCAP; cntS[p] = (u32)cnt;
		if (cap == 0) { large[p] = 0; --n_large; continue; }       /* cannot happen: a large sub-table has keys or a table */
		for (;;) {
			const u64 thr = (cap >> 1) + (cap >> 2);
			if (rem > 0) {
				if (cnt >= thr) { sched[p].push_back({ 2u, (u32)ceil_log2_u64(cap), 0u, 0u }); cap <<= 1; continue; }
				const u64 b = std::min(rem, thr - cnt);
				sched[p].push_back({ 1u, (u32)ceil_log2_u64(cap), (u32)(m[p] - rem), (u32)b });
				cnt += b; rem -= b;
			} else {
				if (trail[p] && cnt >= thr) { sched[p].push_back({ 2u, (u32)ceil_log2_u64(cap), 0u, 0u }); cap <<= 1; }
				break;
			}
		}
		bitsF[p] = (u32)ceil_log2_u64(cap); cntF[p] = (u32)cnt;
		if ((1ull << bitsF[p]) > capm[p]) return fail("replay schedule exceeds the planned capacity");
		n_steps = std::max(n_steps, sched[p].size());
	}
	if (n_large == 0) return 1;
	for (int p = 0; p < P; ++p) if (large[p] && (int)bitsF[p] - yk_r2_seg_log() > 10) return 1;   /* more than 1024 segments per sub-table: not handled */
	/* k_replay: the small sub-tables into the final arena, the first part of the large ones into a side arena behind it */
	const u64 tot_ext = tot + (u64)n_large * std::max<u64>(32, SMALLCAP);
	{
		u64 side = tot;
		for (int p = 0; p < P; ++p) {
			ReplayTask &t = tasks[p];
			t.old_bits = from_empty ? YK_NOCAP : c->h_bits[p];
			t.old_count = cnt0[p]; t.old_off = c->h_off[p];
			t.rec_off = rec_off[p]; t.m = m[p];
			t.init_bits = init_bits ? (*init_bits)[p] : YK_NOCAP;
			t.cap_max_bits = capm[p] ? (u32)ceil_log2_u64(capm[p]) : 0;
			t.dbg = (u32)env_i64("YAKAMD_DBG", 0);
			t.new_off = new_off[p];
			if (large[p]) {
				t.new_off = side; side += std::max<u64>(32, SMALLCAP);
				lp_host[p] = 0;                                       /* the trailing put-call is the schedule's business */
				if (cap0[p] > SMALLCAP) { t.old_bits = YK_NOCAP; t.old_count = 0; t.m = 0; t.init_bits = YK_NOCAP; t.cap_max_bits = 0; }   /* already beyond: loaded straight from the old image */
				else { t.m = m1[p]; t.cap_max_bits = SB; }
			}
		}
	}
	const bool par = env_i64("YAKAMD_PAR_REPLAY", 1) != 0;
	/* k_replay's scratch arrays (ranks, second bitmap, doubling lists: 28 bytes per slot) are indexed by arena offsets.  When every sub-table it
	 * touches is a large one -- an assembly, any pass of a big count -- it only works in the side arena behind the final one, so the arrays
	 * cover that alone and are addressed from `tot` on: at 2 G keys they were 85 GB that nothing touched, more than the pool could keep, and the
	 * hipMalloc / hipFree of them cost 5 s per pass (the kernels of the whole layout stage: 0.28 s) */
	bool only_side = true;
	for (int p = 0; p < P; ++p) if (!large[p] && (m[p] || cap0[p])) only_side = false;
	const u64 scr_lo = only_side ? tot : 0, scr_n = tot_ext - scr_lo;
	if (only_side) {
		/* the scratch pointers handed to k_replay below are shifted by scr_lo: that is only sound while the kernel touches no scratch below `tot`,
		 * i.e. while every task outside the side arena is an empty one, and while bitmap words of the two arenas do not straddle */
		if (tot % 32 != 0) return fail("replay: arena size %llu is not a multiple of 32", (unsigned long long)tot);
		for (int p = 0; p < P; ++p) {
			if (large[p]) continue;
			if (tasks[p].m != 0 || tasks[p].old_count != 0 || cap0[p] != 0) return fail("replay: sub-table %d is not empty but lies outside the side arena", p);
			lp_host[p] = 0;                                          /* no put-call can have hit a sub-table that holds nothing: never let a stray time grow it */
		}
	}
	/* Every sub-table that holds anything is a large one and ends at the capacity the arena reserves for it (no trailing doubling left out): the
	 * two buffers the doublings alternate between are then laid out exactly like the arena, and whichever holds most of the final tables BECOMES the
	 * table image -- the others' tables are copied over, nothing else is (the copy of every slot into a third array was 12 ms and 34 GB beside a
	 * 2 Gb assembly).  k_replay's side arena is then all that `nk` / `nu` hold; they are addressed from `tot` on like its scratch */
	bool inplace = only_side;
	for (int p = 0; p < P && inplace; ++p) if (large[p] && (1ull << bitsF[p]) != std::max<u64>(32, capm[p])) inplace = false;
	const u64 nk_lo = inplace ? tot : 0;
	u64 *nk_al = 0; u32 *nu_al = 0, *img_u = 0;                  /* what was allocated: nk / nu below are shifted by nk_lo; img_u: the image's bitmap when a buffer becomes the image */
	struct GuardNk { std::function<void()> f; ~GuardNk() { f(); } } guard_nk{ [&]() { dfree(nk_al); dfree(nu_al); dfree(img_u); nk = 0; nu = 0; } };
	if ((par && dmalloc(&sp, 2 * scr_n)) || dmalloc(&nk_al, tot_ext - nk_lo) || dmalloc(&nu_al, (tot_ext - nk_lo) / 32 + 1) || dmalloc(&su, scr_n / 32 + 1) || dmalloc(&so, scr_n) ||
	    dmalloc(&d_tasks, P) || dmalloc(&d_ob, P) || dmalloc(&d_oc, P)) return -1;
	nk = nk_al - nk_lo; nu = nu_al - nk_lo / 32;
	if (inplace) {
		HIPCK(hipMemsetAsync(nk_al, 0xff, (tot_ext - tot) * 8, c->st));
		HIPCK(hipMemsetAsync(nu_al, 0, ((tot_ext - tot) / 32 + 1) * 4, c->st));
	} else if (only_side) {
		/* k_r2_publish writes every slot and every bitmap word of a large sub-table: only the side arena and the (empty, 32-slot) regions of
		 * the other sub-tables need the empty pattern -- not 8 bytes per slot of the whole arena (1 Gb assembly: 2.9 ms) */
		HIPCK(hipMemsetAsync(nk + tot, 0xff, (tot_ext - tot) * 8, c->st));
		HIPCK(hipMemsetAsync(nu + tot / 32, 0, ((tot_ext - tot) / 32 + 1) * 4, c->st));
		for (int p = 0; p < P;) {
			if (large[p]) { ++p; continue; }
			int q = p;
			while (q < P && !large[q]) ++q;                          /* a run of sub-tables without a large table: contiguous in the arena */
			const u64 a = new_off[p], b = q < P ? new_off[q] : tot;
			HIPCK(hipMemsetAsync(nk + a, 0xff, (b - a) * 8, c->st));
			HIPCK(hipMemsetAsync(nu + a / 32, 0, (b - a) / 32 * 4, c->st));
			p = q;
		}
	} else {
		HIPCK(hipMemsetAsync(nk, 0xff, tot_ext * 8, c->st));
		HIPCK(hipMemsetAsync(nu, 0, (tot_ext / 32 + 1) * 4, c->st));
	}
	HIPCK(hipMemcpyAsync(d_tasks, tasks.data(), P * sizeof(ReplayTask), hipMemcpyHostToDevice, c->st));
	if (d_lastput) HIPCK(hipMemcpyAsync(d_lp2, lp_host.data(), P * 8, hipMemcpyHostToDevice, c->st));
	legacy_replay_launch(c, tasks, d_tasks, nk, nu, su - scr_lo / 32, so - scr_lo, sp ? sp - 2 * scr_lo : 0, d_rec_kc, d_rec_t, d_lastput ? d_lp2 : 0, d_ob, d_oc);
	std::vector<u32> ob(P), oc(P);
	HIPCK(hipMemcpyAsync(ob.data(), d_ob, P * 4, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipMemcpyAsync(oc.data(), d_oc, P * 4, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	dfree(sp); dfree(su); dfree(so);
	/* buffers of the large sub-tables */
	std::vector<R2Tab> tabs(P);
	std::vector<u32> seg0(P, 0);
	const int SEGLOG = yk_r2_seg_log();
	u64 tot2 = 0, nseg_tot = 0; u32 bmaxF = 0, bmaxS = 0;
	for (int p = 0; p < P; ++p) {
		tabs[p].off = inplace ? new_off[p] : tot2; tabs[p].rec_off = rec_off[p];
		if (!large[p]) continue;
		if (cap0[p] <= SMALLCAP && (ob[p] != bitsS[p] || oc[p] != cntS[p])) return fail("replay: sub-table %d left k_replay with 2^%u slots / %u keys, the schedule says 2^%u / %u", p, ob[p], oc[p], bitsS[p], cntS[p]);
		tot2 += 1ull << bitsF[p];
		seg0[p] = (u32)nseg_tot;
		nseg_tot += (bitsF[p] > (u32)SEGLOG ? 1ull << (bitsF[p] - SEGLOG) : 1) + 1;
		bmaxF = std::max(bmaxF, bitsF[p]); bmaxS = std::max(bmaxS, bitsS[p]);
	}
	std::vector<R2Act> acts(std::max<size_t>(1, n_steps) * P);
	memset(acts.data(), 0, acts.size() * sizeof(R2Act));
	std::vector<R2Load> ld(P); std::vector<R2Pub> pub(P);
	u64 side = tot;
	for (int p = 0; p < P; ++p) {
		ld[p].bits = YK_NOCAP; ld[p].src_off = 0; ld[p].from_src = 0; ld[p].dst = 0; ld[p].pad = 0;
		pub[p].bits = YK_NOCAP; pub[p].new_off = new_off[p]; pub[p].src = 0;
		if (!large[p]) continue;
		ld[p].bits = bitsS[p];
		if (cap0[p] > SMALLCAP) { const bool old = !from_empty && c->h_bits[p] != YK_NOCAP; ld[p].from_src = old ? 2 : 0; ld[p].src_off = old ? c->h_off[p] : 0; }
		else { ld[p].from_src = 1; ld[p].src_off = side; }
		side += std::max<u64>(32, SMALLCAP);
		u32 src = 0;
		for (size_t k = 0; k < sched[p].size(); ++k) {
			R2Act &a = acts[k * P + p];
			a.kind = sched[p][k].kind; a.bits = sched[p][k].bits; a.i0 = sched[p][k].i0; a.batch = sched[p][k].batch; a.src = src; a.seg0 = seg0[p];
			if (a.kind == 2) src ^= 1;
		}
		pub[p].bits = bitsF[p]; pub[p].src = src;
	}
	/* the buffer that becomes the image, known before anything runs (the schedule is simulated): a sub-table that ends there with a placement gets
	 * its "used" bits from that step's kernels (R2Act.pad0) and needs no pass of k_r2_publish */
	bool img_is1 = false;
	std::vector<char> pub_needed(P, 1);
	if (inplace) {
		u64 in1 = 0, in0 = 0;
		for (int p = 0; p < P; ++p) if (large[p]) (pub[p].src ? in1 : in0) += 1ull << bitsF[p];
		img_is1 = in1 > in0;
		if (dmalloc(&img_u, tot / 32 + 1)) return -1;
		for (int p = 0; p < P; ++p) {
				if (!large[p] || sched[p].empty() || sched[p].back().kind != 1 || (pub[p].src != 0) != img_is1) continue;
				acts[(sched[p].size() - 1) * P + p].pad0 = 1;
				pub_needed[p] = 0;
			}
	}
	if (inplace) tot2 = tot;                                     /* the buffers are arenas */
	const u32 spill_cap = (u32)std::min<u64>(1u << 28, std::max<u64>(1u << 20, tot2 / 16));   /* also the list of long runs of a doubling round */
	u64 n_keys = 0;
	for (int p = 0; p < P; ++p) n_keys = std::max(n_keys, rec_off[p] + m[p]);
	if (dmalloc(&K0, tot2) || dmalloc(&K1, tot2) || dmalloc(&TAG, tot2 / 2 + 1) || dmalloc(&OCC, tot2 / 16 + (size_t)P + 64) || dmalloc(&USED, tot2 / 32 + 64) || dmalloc(&d_tabs, P) || dmalloc(&d_acts, acts.size()) || dmalloc(&d_ld, P) || dmalloc(&d_pub, P) ||
	    dmalloc(&pk, n_keys) || dmalloc(&pr, n_keys) || dmalloc(&segst, nseg_tot + 1) || dmalloc(&head, (size_t)nseg_tot * yk_r2_head()) || dmalloc(&spill, spill_cap) || dmalloc(&Fc, 4 * (size_t)P) || dmalloc(&misc, 4)) return -1;
	/* few large sub-tables (a shard): the keys of a stage are grouped by G workgroups per sub-table instead of one (YAKAMD_R2_PPART_G: tests) */
	int ppG = (int)std::min<int64_t>(16, std::max<int64_t>(1, env_i64("YAKAMD_R2_PPART_G", n_large <= 512 ? 1024 / std::max<u32>(1, n_large) : 1)));
	if ((size_t)P * ppG > (64u << 10)) ppG = 1;                   /* (the counters are indexed by sub-table: 4 KB per sub-table and share) */
	if (ppG > 1 && dmalloc(&pcnt, (size_t)P * ppG * 1024)) return -1;
	HIPCK(hipMemcpyAsync(d_tabs, tabs.data(), P * sizeof(R2Tab), hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemcpyAsync(d_acts, acts.data(), acts.size() * sizeof(R2Act), hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemcpyAsync(d_ld, ld.data(), P * sizeof(R2Load), hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemcpyAsync(d_pub, pub.data(), P * sizeof(R2Pub), hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemsetAsync(misc, 0, 16, c->st));
	u32 *d_fail = misc, *d_nspill = misc + 1;
	yk_r2_load(d_tabs, d_ld, P, bmaxS, nk, c->d_keys, K0, K1, USED, c->st);
	const bool prof = env_i64("YAKAMD_VERBOSE", 0) > 1;
	auto lap = [&](const char *what, size_t k, u32 bits, double *t0) {
		if (!prof) return;
		hipStreamSynchronize(c->st);
		const double t1 = now_ms();
		fprintf(stderr, "[yak_amd] replay2 step %zu (2^%u): %s %.3f ms\n", k, bits, what, t1 - *t0);
		*t0 = t1;
	};
	double tl = now_ms();
	lap("k_replay part + load", 0, bmaxS, &tl);
	for (size_t k = 0; k < n_steps; ++k) {
		u32 bd = 0, bp = 0; bool any_d = false, any_p = false;
		int p_lo = P, p_hi = 0;                                      /* the sub-tables that place in this step: a shard's are a contiguous range of the P */
		for (int p = 0; p < P; ++p) {
			const R2Act &a = acts[k * P + p];
			if (a.kind == 2) { any_d = true; bd = std::max(bd, a.bits); }
			else if (a.kind == 1) { any_p = true; bp = std::max(bp, a.bits); p_lo = std::min(p_lo, p); p_hi = p + 1; }
		}
		const R2Act *da = d_acts + k * P;
		if (any_d) {
			yk_r2_binit(d_tabs, da, P, bd, OCC, USED, c->st);
			lap("binit", k, bd, &tl);
			int n_dbl = 0;
			for (int p = 0; p < P; ++p) n_dbl += acts[k * P + p].kind == 2;
			yk_r2_dsmall(d_tabs, da, P, K0, K1, TAG, OCC, USED, Fc, Fc + 2 * P, d_fail, c->st);
			lap("dsmall", k, bd, &tl);
			/* the rounds from there on in one launch: a workgroup per sub-table walks its rounds behind workgroup barriers.  A sub-table that does
			 * not reach its end raises `fail` (read once, after the last step: whatever the later steps then do is thrown away with the buffers) */
			yk_r2_double(d_tabs, da, P, n_dbl, K0, K1, TAG, OCC, USED, Fc, Fc + P, d_fail, c->st);
			lap("double (fused rounds)", k, bd, &tl);
		}
		if (any_p) { yk_r2_place(d_tabs, da, P, p_lo, p_hi - p_lo, bp, K0, K1, d_rec_kc, pk, pr, segst, head, spill, d_nspill, spill_cap, d_fail, img_u, USED, pcnt, ppG, c->st); lap("place", k, bp, &tl); }
	}
	u64 *img_k = 0;                                               /* the new image, once it is certain */
	if (inplace) {
		u64 *A = img_is1 ? K1 : K0;
		/* the regions of the sub-tables that hold nothing: empty pattern, no bit */
		for (int p = 0; p < P;) {
			if (large[p]) { ++p; continue; }
			int q = p;
			while (q < P && !large[q]) ++q;
			const u64 a = new_off[p], b = q < P ? new_off[q] : tot;
			HIPCK(hipMemsetAsync(A + a, 0xff, (b - a) * 8, c->st));
			HIPCK(hipMemsetAsync(img_u + a / 32, 0, (b - a) / 32 * 4, c->st));
			p = q;
		}
		for (int p = 0; p < P; ++p) { pub[p].new_off = tabs[p].off; if (!pub_needed[p]) pub[p].bits = YK_NOCAP; }
		HIPCK(hipMemcpyAsync(d_pub, pub.data(), P * sizeof(R2Pub), hipMemcpyHostToDevice, c->st));
		yk_r2_publish(d_tabs, d_pub, P, bmaxF, K0, K1, A, img_u, c->st);   /* a table already in A only gets its bitmap */
		img_k = A;
	} else yk_r2_publish(d_tabs, d_pub, P, bmaxF, K0, K1, nk, nu, c->st);
	lap("publish", n_steps, bmaxF, &tl);
	u32 h_fail = 0;
	HIPCK(hipMemcpyAsync(&h_fail, d_fail, 4, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	if (h_fail) {
		if (env_i64("YAKAMD_VERBOSE", 0)) fprintf(stderr, "[yak_amd] streaming replay refused (code %u): falling back to k_replay\n", h_fail);
		++g_r2_refused;
		return 1;
	}
	++g_r2_used;
	for (int p = 0; p < P; ++p) {
		if (large[p]) { c->h_bits[p] = bitsF[p]; c->h_count[p] = cntF[p]; }
		else { c->h_bits[p] = ob[p]; c->h_count[p] = oc[p]; }
	}
	dfree(c->d_keys); dfree(c->d_used); dfree(c->d_delta);
	if (inplace) {
		c->d_keys = img_k; c->d_used = img_u; img_u = 0;
		if (img_k == K0) K0 = 0; else K1 = 0;                     /* the guard releases the other one */
	} else { c->d_keys = nk_al; c->d_used = nu_al; nk_al = 0; nu_al = 0; }
	c->n_slots = tot;
	c->h_off = new_off;
	HIPCK(hipMemcpyAsync(c->d_bits, c->h_bits.data(), P * 4, hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemcpyAsync(c->d_off, c->h_off.data(), P * 8, hipMemcpyHostToDevice, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	c->img_keys_total = 0;
	for (int p = 0; p < P; ++p) c->img_keys_total += c->h_count[p];
	c->host_valid = false;
	return 0;
}

int yk_run_replay(yakamd_ctx *c, const std::vector<u32> &m, const u64 *d_seg_off, const u64 *d_rec_kc, const u64 *d_rec_t,
                  const u64 *d_lastput, const std::vector<u32> *init_bits, bool from_empty, const std::vector<u64> *rec_off)
{
	{
		const int r2 = run_replay_v2(c, m, d_rec_kc, d_rec_t, d_lastput, init_bits, from_empty, rec_off);
		if (r2 <= 0) return r2;
	}
	const int P = c->P;
	std::vector<ReplayTask> tasks(P);
	std::vector<u64> new_off(P);
	u64 tot = 0, rec = 0;
	for (int p = 0; p < P; ++p) {
		ReplayTask &t = tasks[p];
		t.old_bits = from_empty ? YK_NOCAP : c->h_bits[p];
		t.old_count = from_empty ? 0 : c->h_count[p];
		t.old_off = c->h_off[p];
		t.rec_off = rec_off ? (*rec_off)[p] : rec; t.m = m[p]; rec += m[p];
		t.init_bits = init_bits ? (*init_bits)[p] : YK_NOCAP;
		u32 cap0 = t.old_bits == YK_NOCAP ? 0 : 1u << t.old_bits;
		if (cap0 == 0 && t.init_bits != YK_NOCAP) cap0 = 1u << t.init_bits;
		const u32 capm = plan_cap(cap0, t.old_count, t.m, d_lastput != 0);
		t.cap_max_bits = capm ? (u32)ceil_log2_u64(capm) : 0;
		t.dbg = (u32)env_i64("YAKAMD_DBG", 0);
		new_off[p] = tot; t.new_off = tot;
		tot += std::max<u64>(32, capm);
	}
	(void)d_seg_off;
	u64 *nk = 0, *sp = 0; u32 *nu = 0, *su = 0, *so = 0, *d_ob = 0, *d_oc = 0;
	ReplayTask *d_tasks = 0;
	struct Guard { std::function<void()> f; ~Guard() { f(); } } guard{ [&]() {   /* nk / nu are handed to the context on success (set to 0 there) */
		dfree(su); dfree(so); dfree(sp); dfree(d_tasks); dfree(d_ob); dfree(d_oc); dfree(nk); dfree(nu);
	} };
	const bool par = env_i64("YAKAMD_PAR_REPLAY", 1) != 0;
	if ((par && dmalloc(&sp, 2 * tot)) || dmalloc(&nk, tot) || dmalloc(&nu, tot / 32) || dmalloc(&su, tot / 32) || dmalloc(&so, tot) ||
	    dmalloc(&d_tasks, P) || dmalloc(&d_ob, P) || dmalloc(&d_oc, P)) return -1;
	HIPCK(hipMemsetAsync(nk, 0xff, tot * 8, c->st));
	HIPCK(hipMemsetAsync(nu, 0, tot / 8, c->st));
	HIPCK(hipMemcpyAsync(d_tasks, tasks.data(), P * sizeof(ReplayTask), hipMemcpyHostToDevice, c->st));
	/* few, large sub-tables (a shard of a multi-GPU job): more lanes per sub-table */
	const int n_active = c->phi - c->plo;
	/* owner ranks of the placement stages in LDS: 32-bit up to lds_words slots, 16-bit up to twice that */
	u32 cap_top = 0;
	for (int p = 0; p < P; ++p) if (tasks[p].m) cap_top = std::max(cap_top, 1u << tasks[p].cap_max_bits);
	u32 lds_words = std::min<u32>(cap_top, (u32)env_i64("YAKAMD_REPLAY_LDS", 16384));   /* 64 KB: two workgroups per CU (measured 18.5 ms against 20.5 with 128 KB); 0: owner ranks in global scratch */
	int n_thr = n_active <= 256 ? 1024 : n_active <= 512 ? 512 : 256;
	if (lds_words * 4 >= 96 * 1024) n_thr = 1024; else if (lds_words * 4 >= 48 * 1024) n_thr = std::max(n_thr, 512);
	n_thr = (int)env_i64("YAKAMD_REPLAY_THREADS", n_thr);
	yk_launch_replay(d_tasks, P, n_thr, c->d_keys, c->d_used, nk, nu, su, so, sp, d_rec_kc, d_rec_t, d_lastput, d_ob, d_oc, lds_words, c->st);
	if (env_i64("YAKAMD_DBG", 0) & 32) {
		HIPCK(hipStreamSynchronize(c->st));
		u64 pr[8]; yk_replay_prof(pr);
		fprintf(stderr, "[yak_amd] replay block 0 (100 MHz ticks): double<32K %llu, double>=32K %llu, place<32K %llu, place>=32K %llu, publish %llu | par doubling: setup+base %llu, rounds %llu, verify+commit %llu\n",
		        (unsigned long long)pr[0], (unsigned long long)pr[1], (unsigned long long)pr[2], (unsigned long long)pr[3], (unsigned long long)pr[4],
		        (unsigned long long)pr[5], (unsigned long long)pr[6], (unsigned long long)pr[7]);
	}
	HIPCK(hipMemcpyAsync(c->h_bits.data(), d_ob, P * 4, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipMemcpyAsync(c->h_count.data(), d_oc, P * 4, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	dfree(c->d_keys); dfree(c->d_used); dfree(c->d_delta);
	c->d_keys = nk; c->d_used = nu; c->n_slots = tot;
	nk = 0; nu = 0;
	c->h_off = new_off;
	HIPCK(hipMemcpyAsync(c->d_bits, c->h_bits.data(), P * 4, hipMemcpyHostToDevice, c->st));
	HIPCK(hipMemcpyAsync(c->d_off, c->h_off.data(), P * 8, hipMemcpyHostToDevice, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	c->img_keys_total = 0;
	for (int p = 0; p < P; ++p) c->img_keys_total += c->h_count[p];
	c->host_valid = false;
	return 0;
}

void yk_replay_counters(u32 *used, u32 *refused) { *used = g_r2_used; *refused = g_r2_refused; }
