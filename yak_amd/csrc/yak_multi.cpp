/*
 * yak_multi.cpp -- several GPUs (or several sweeps over prefix ranges on one) behind yak_count(), and the same job on input that is already in HBM
 * (yakamd_count_multi_dev): SURVEY 8e, reference count.c:129-143.
 */
#include "yak_host.h"

/* ------------------------------------------------------------------------------------------
 * Several GPUs behind yak_count() (replaces the kt_for over prefixes, count.c:129-143).
 * YAKAMD_GPUS = N (a divisor of 1 << pre): GPU r owns the contiguous prefixes [r P / N, (r + 1) P / N) --
 * table, filters and all.  The input is dealt to the GPUs in chunks of YAKAMD_MGPU_CHUNK bytes of sequence:
 * chunk j goes to GPU j % N, which extracts and groups its k-mers by prefix (yakamd_partition_dev); one
 * exchange per round of N chunks then moves every record to the owner of its prefix -- RCCL
 * (ncclGroupStart + ncclSend / ncclRecv pairs over xGMI, one communicator per GPU from ncclCommInitAll),
 * or plain device copies when two ranks share a GPU (YAKAMD_GPU_LIST=0,0: one-GPU test rigs); the owner
 * feeds the slices in chunk order, which is the stream order of the file, so the N-GPU bytes are the
 * 1-GPU bytes.  librccl is opened only when a job asks for several distinct GPUs.
 * ------------------------------------------------------------------------------------------ */
bool env_fast_default() { return yk_knob("YAKAMD_FAST", 1) != 0; }   /* the exclusive-ownership path (the only one that takes tagged records) is on */

struct RcclApi {
	void *lib;
	ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*);
	ncclResult_t (*CommDestroy)(ncclComm_t);
	ncclResult_t (*GroupStart)(void);
	ncclResult_t (*GroupEnd)(void);
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
	const char *(*GetErrorString)(ncclResult_t);
};
static bool rccl_open(RcclApi *R)
{
	memset(R, 0, sizeof(*R));
	R->lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
	if (!R->lib) R->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
	if (!R->lib) return false;
#define YK_SYM(f, n) *(void**)&R->f = dlsym(R->lib, n)
	YK_SYM(CommInitAll, "ncclCommInitAll"); YK_SYM(CommDestroy, "ncclCommDestroy"); YK_SYM(GroupStart, "ncclGroupStart"); YK_SYM(GroupEnd, "ncclGroupEnd");
	YK_SYM(Send, "ncclSend"); YK_SYM(Recv, "ncclRecv"); YK_SYM(GetErrorString, "ncclGetErrorString");
#undef YK_SYM
	return R->CommInitAll && R->CommDestroy && R->GroupStart && R->GroupEnd && R->Send && R->Recv;
}

/* Test rig (YAKAMD_MGPU_LOOPBACK, a test switch): the same seven entry points served by copies on the devices of this process, so that the whole grouped
 * send / recv call pattern of a round -- who sends what to whom, in which order, on which stream -- runs where no second GPU (or no xGMI) is: every
 * ncclRecv of a group must find the ncclSend of its peer with the SAME count, in posting order per (sender, receiver) pair, exactly what the real
 * library requires (a mismatch there is a hang; here it is ncclInvalidArgument).  The copy of a pair is ordered behind the sender's stream and ahead of
 * both streams' later work through events, as a real transfer is.  YAKAMD_MGPU_LOOPBACK_FAIL = r (r >= 1) makes the r-th ncclGroupEnd of the process
 * fail without moving anything: the branch that repeats a round as peer copies. */
struct LoopComm { int rank, n, dev; };
struct LoopOp { const void *src; void *dst; size_t cnt; int from, to, dev; hipStream_t st; };
static std::mutex g_loop_mu;
static std::vector<LoopOp> g_loop_send, g_loop_recv;
static int g_loop_groups = 0;
static ncclResult_t loop_init_all(ncclComm_t *c, int n, const int *dev) { for (int i = 0; i < n; ++i) { LoopComm *l = new LoopComm; l->rank = i; l->n = n; l->dev = dev[i]; c[i] = (ncclComm_t)l; } return ncclSuccess; }
static ncclResult_t loop_destroy(ncclComm_t c) { delete (LoopComm*)c; return ncclSuccess; }
static ncclResult_t loop_group_start(void) { std::lock_guard<std::mutex> lk(g_loop_mu); g_loop_send.clear(); g_loop_recv.clear(); return ncclSuccess; }
static ncclResult_t loop_send(const void *p, size_t cnt, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st)
{
	LoopComm *l = (LoopComm*)c;
	if (t != ncclUint64 || peer < 0 || peer >= l->n || peer == l->rank) return ncclInvalidArgument;
	std::lock_guard<std::mutex> lk(g_loop_mu);
	g_loop_send.push_back(LoopOp{ p, 0, cnt, l->rank, peer, l->dev, st });
	return ncclSuccess;
}
static ncclResult_t loop_recv(void *p, size_t cnt, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st)
{
	LoopComm *l = (LoopComm*)c;
	if (t != ncclUint64 || peer < 0 || peer >= l->n || peer == l->rank) return ncclInvalidArgument;
	std::lock_guard<std::mutex> lk(g_loop_mu);
	g_loop_recv.push_back(LoopOp{ 0, p, cnt, peer, l->rank, l->dev, st });
	return ncclSuccess;
}
static ncclResult_t loop_group_end(void)
{
	std::lock_guard<std::mutex> lk(g_loop_mu);
	++g_loop_groups;
	if (yk_knob("YAKAMD_MGPU_LOOPBACK_FAIL", 0) == g_loop_groups) { g_loop_send.clear(); g_loop_recv.clear(); return ncclSystemError; }
	if (g_loop_send.size() != g_loop_recv.size()) return ncclInvalidArgument;
	std::vector<char> taken(g_loop_send.size(), 0);
	ncclResult_t res = ncclSuccess;
	for (const LoopOp &r : g_loop_recv) {
		size_t i = 0;
		while (i < g_loop_send.size() && (taken[i] || g_loop_send[i].from != r.from || g_loop_send[i].to != r.to)) ++i;    /* the pair's sends in posting order */
		if (i == g_loop_send.size() || g_loop_send[i].cnt != r.cnt) { res = ncclInvalidArgument; break; }
		const LoopOp &sn = g_loop_send[i];
		taken[i] = 1;
		hipEvent_t posted = 0, moved = 0;
		if (hipSetDevice(sn.dev) != hipSuccess || hipEventCreateWithFlags(&posted, hipEventDisableTiming) != hipSuccess || hipEventRecord(posted, sn.st) != hipSuccess) { res = ncclUnhandledCudaError; break; }
		if (hipSetDevice(r.dev) != hipSuccess || hipStreamWaitEvent(r.st, posted, 0) != hipSuccess
		    || hipMemcpyPeerAsync(r.dst, r.dev, sn.src, sn.dev, r.cnt * 8, r.st) != hipSuccess
		    || hipEventCreateWithFlags(&moved, hipEventDisableTiming) != hipSuccess || hipEventRecord(moved, r.st) != hipSuccess
		    || hipStreamWaitEvent(sn.st, moved, 0) != hipSuccess) res = ncclUnhandledCudaError;          /* the send buffer is the sender's again once its stream has passed this point */
		if (posted) (void)hipEventDestroy(posted);                  /* (released by the runtime when the work that refers to them is done) */
		if (moved) (void)hipEventDestroy(moved);
		if (res != ncclSuccess) break;
	}
	g_loop_send.clear(); g_loop_recv.clear();
	return res;
}
static const char *loop_error_string(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidArgument ? "loopback rig: a receive without the matching send" : "loopback rig: error"; }
static void loop_open(RcclApi *R)
{
	memset(R, 0, sizeof(*R));
	R->CommInitAll = loop_init_all; R->CommDestroy = loop_destroy; R->GroupStart = loop_group_start; R->GroupEnd = loop_group_end;
	R->Send = loop_send; R->Recv = loop_recv; R->GetErrorString = loop_error_string;
}

/* The communicators of a device list are made once per process and kept: ncclCommInitAll over 8 GPUs takes of the order of a second and the
 * first grouped send / recv of a pair sets its channels up, while a counting job of this size takes less than a second -- a process that counts
 * again (the benchmark's steps, the second pass of the filtered protocol, a server) finds them here.  One job at a time holds them; a job that
 * finds them taken makes its own and destroys them when it is done. */
struct CommCache {
	std::mutex mu;
	bool opened, have_lib, busy;
	RcclApi R;
	std::vector<int> devs;
	std::vector<ncclComm_t> comm;
	CommCache() : opened(false), have_lib(false), busy(false) {}
};
static CommCache g_cc;

/* Ranks own prefix ranges; chunks of the input live in SLOTS, one per distinct device (ranks that share a device -- the sweeps of one device
 * posing as several -- share its chunk, its partition and its buffers: the owner's slice of a chunk on its own device is fed where it lies).
 * Two sets of slot buffers: set x is being partitioned, exchanged and fed by a worker thread while the reader fills set 1 - x. */
struct MultiJob {
	int N, P, S;                                               /* ranks, sub-tables, slots */
	std::vector<int> dev, sdev, slot_of;                       /* device of rank r; device of slot s; slot of rank r */
	std::vector<hipStream_t> st, cp;                           /* per slot: exchange stream; copy stream of the reader (non-blocking: the fill of the next set must not wait for the kernels of this one) */
	bool use_rccl, loopback;                                   /* loopback: the test rig above stands in for the collective library */
	RcclApi R;
	std::vector<ncclComm_t> comm;                              /* per slot */
	bool comm_cached;                                          /* they are g_cc's: handed back, not destroyed */
	std::vector<uint8_t*> d_base[2];                           /* [set][slot]: chunk of sequence */
	std::vector<uint64_t*> d_send[2], d_recv[2];               /* [set][slot]: records grouped by prefix / slices received from the other slots */
	int64_t chunk, send_words, recv_words;
	bool ext_base;                                             /* d_base points at the caller's device buffers (yakamd_count_multi_dev) */
};

/* no filter + a plain file of more than YAKAMD_AUTO_SWEEP_GB (2.5) GB: nearly every k-mer instance may be a key of its own (an assembly),
 * and one pass holds ~70 bytes per selected key at its peak -- such inputs are counted as N ranks on one device, i.e. in N sweeps over
 * prefix ranges (N so that a sweep sees at most ~2.8 G positions of its own: 5 Gb in 2 sweeps, 2.4 s on a device whose memory has been in use
 * before, 2.4 s in 4; round 3 needed 4 -- a rank of 2 held a third copy of its table and 4 bytes of pending counts per slot while its layout
 * was replayed).  YAKAMD_GPUS set to anything switches the rule off.
 *
 * What the memory costs (round 6).  `yak count` is one job per process (main.c:53-61), and a process pays the driver ~30 ms for every GB of device
 * memory beyond the first ~150 (profiles/r06_mb_vmm5.txt; from any number of threads): the 5 Gb assembly in 2 sweeps obtains 239 GB and spends 2.7 of
 * its 4.4 s there, in 8 sweeps it peaks at 103 GB, obtains 150 and takes 1.96 s -- 1.77 s once the memory is the process's own, against 1.60-1.68 in 2
 * (profiles/r06_experiments.txt e11).  So a process that does not hold the memory of the plan above yet takes more sweeps: the smallest N whose
 * estimated peak -- 14 bytes per input byte for the table of an assembly + 56 per byte of a sweep's share -- stays within YAKAMD_COLD_GB (110; 0 = the
 * rule above alone).  That also sends files between 1.6 and 2.5 GB through 2 sweeps on a cold device.  A process whose pool already holds what the
 * smaller N needs (an earlier job obtained it) keeps the smaller N. */
static int g_last_sweeps = 1;
extern "C" int yakamd_last_sweeps(void) { return g_last_sweeps; }   /* ranks (= sweeps on one device) of this process's last yak_count() */
int auto_sweeps(const yak_copt_t *opt, const char *fn)
{
	if (fn == 0 || strcmp(fn, "-") == 0 || opt->bf_shift > opt->pre) return 1;
	const char *g = getenv("YAKAMD_AUTO_SWEEP_GB");
	const double lim = (g ? atof(g) : 2.5) * 1e9;
	if (lim <= 0) return 1;
	struct stat sb;
	if (stat(fn, &sb) != 0 || !S_ISREG(sb.st_mode)) return 1;
	unsigned char m[2] = { 0, 0 };
	const int f = ::open(fn, O_RDONLY);
	if (f < 0) return 1;
	const bool gz = ::read(f, m, 2) == 2 && m[0] == 0x1f && m[1] == 0x8b;
	::close(f);
	if (gz) return 1;                                           /* compressed: the size says little; the knob is there */
	const int P = 1 << opt->pre;
	const double sz = (double)sb.st_size;
	int N = 1;
	if (sz > lim) { N = 2; while (N < 16 && sz / N > 2.8e9) N <<= 1; }
	const double bases = m[0] == '@' ? sz * 0.5 : sz;           /* a FASTQ record spends half of its bytes on the quality line */
	const char *cg = getenv("YAKAMD_COLD_GB");
	const double cold = (cg ? atof(cg) : 110.0) * 1e9;
	auto peak = [&](int n) { return bases * (14.0 + 56.0 / n); };
	if (cold > 0 && peak(N) > cold) {
		int nd = 0, d = 0;
		if (hipGetDeviceCount(&nd) == hipSuccess && nd > 0) { const char *dv = getenv("YAKAMD_DEVICE"), *lr = getenv("LOCAL_RANK"); d = (dv ? atoi(dv) : lr ? atoi(lr) : 0) % nd; }
		if ((double)yk_pool_held_bytes(d) < peak(N))            /* the process does not own that memory yet */
			while (N < 16 && P % (2 * N) == 0 && peak(N) > cold) N <<= 1;
	}
	return N > 1 && P % N ? 1 : N;
}

int multi_gpus(const yak_copt_t *opt, std::vector<int> *dev, const char *fn)
{
	const char *e = getenv("YAKAMD_GPUS");
	if (!e) {
		const int S = auto_sweeps(opt, fn);
		g_last_sweeps = 1;
		if (S <= 1) return 1;
		int nd = 0;
		if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1) return 1;
		const char *dv = getenv("YAKAMD_DEVICE"), *lr = getenv("LOCAL_RANK");
		const int d = (dv ? atoi(dv) : lr ? atoi(lr) : 0) % nd;
		dev->assign(S, d);
		fprintf(stderr, "[M::yak_count] %s: no filter and a large plain file: counting in %d sweeps over prefix ranges on device %d (YAKAMD_GPUS / YAKAMD_AUTO_SWEEP_GB / YAKAMD_COLD_GB change that)\n", fn, S, d);
		g_last_sweeps = S;
		return S;
	}
	const int N = atoi(e);
	if (N <= 1) return 1;
	int nd = 0;
	if (hipGetDeviceCount(&nd) != hipSuccess || nd < 1) return 1;
	if ((1 << opt->pre) % N) { fprintf(stderr, "[W::yak_count] YAKAMD_GPUS=%d does not divide the %d sub-tables: counting on one GPU\n", N, 1 << opt->pre); return 1; }
	dev->clear();
	if (const char *l = getenv("YAKAMD_GPU_LIST")) { for (const char *q = l; *q; ) { dev->push_back(atoi(q) % nd); while (*q && *q != ',') ++q; if (*q) ++q; } }
	for (int r = (int)dev->size(); r < N; ++r) dev->push_back(r % nd);
	dev->resize(N);
	g_last_sweeps = N;
	return N;
}

static bool multi_open(MultiJob *J, int N, int P, const std::vector<int> &dev, int64_t chunk_dev = 0, bool tagged_only = false)   /* chunk_dev > 0: the chunks are the caller's device buffers of at most that many bytes */
{
	J->N = N; J->P = P; J->dev = dev;
	J->sdev.clear(); J->slot_of.assign(N, 0);
	/* test switch: every rank a slot of its own although ranks share a device -- chunk, send and receive buffers, streams and staging events per slot, and
	 * an exchange between slots (copies on the one device stand in for the wire): everything a box with several GPUs runs, on a box with one */
	const bool own_slots = yk_knob("YAKAMD_MGPU_SLOT_PER_RANK", 0) != 0;
	bool dup_dev = false;
	for (int r = 0; r < N; ++r) {
		int s = -1;
		for (size_t q = 0; q < J->sdev.size(); ++q) if (J->sdev[q] == dev[r]) { if (own_slots) dup_dev = true; else s = (int)q; }
		if (s < 0) { s = (int)J->sdev.size(); J->sdev.push_back(dev[r]); }
		J->slot_of[r] = s;
	}
	const int S = J->S = (int)J->sdev.size();
	J->st.assign(S, 0); J->cp.assign(S, 0);
	for (int x = 0; x < 2; ++x) { J->d_base[x].assign(S, 0); J->d_send[x].assign(S, 0); J->d_recv[x].assign(S, 0); }
	const char *c = getenv("YAKAMD_MGPU_CHUNK");
	J->chunk = c && atoll(c) > 0 ? atoll(c) : (int64_t)1 << 28;
	if (chunk_dev > 0) J->chunk = chunk_dev;
	J->chunk = (J->chunk + 4095) & ~(int64_t)4095;
	J->ext_base = chunk_dev > 0;
	J->send_words = (tagged_only ? 1 : 2) * J->chunk;          /* one 16-byte record per position at most (8-byte tagged records use half of it) */
	/* a slot receives, for the ranks it hosts, their share of the S - 1 other chunks: (ranks here / N) each on average; refused beyond 1.5 x that */
	int most = 0;
	for (int s = 0; s < S; ++s) { int n_here = 0; for (int r = 0; r < N; ++r) n_here += J->slot_of[r] == s; most = std::max(most, n_here); }
	J->recv_words = S > 1 ? (int64_t)((double)J->send_words * (S - 1) * most / N * 1.5) + 4096 * N : 0;
	const bool loopback = S > 1 && yk_knob("YAKAMD_MGPU_LOOPBACK", 0) != 0;   /* test switch: the collective calls served inside the process (above) */
	J->loopback = loopback;
	J->use_rccl = S > 1 && !yk_knob("YAKAMD_MGPU_NO_RCCL", 0) && (loopback || !dup_dev);   /* (the real library refuses a device twice in one list) */
	if (S < N) fprintf(stderr, "[M::yak_count] %d ranks on %d device%s: ranks that share a device share its chunks and take turns (their slices are fed where they lie)%s\n",
	                   N, S, S > 1 ? "s" : "", S == 1 ? "; nothing is exchanged" : "");
	J->comm_cached = false;
	if (J->use_rccl && loopback) {
		J->comm.assign(S, 0);
		loop_open(&J->R);
		J->R.CommInitAll(J->comm.data(), S, J->sdev.data());
	} else if (J->use_rccl) {
		J->comm.assign(S, 0);
		std::lock_guard<std::mutex> lk(g_cc.mu);
		if (!g_cc.opened) { g_cc.have_lib = rccl_open(&g_cc.R); g_cc.opened = true; }
		J->R = g_cc.R;
		if (!g_cc.have_lib) { fprintf(stderr, "[W::yak_count] librccl.so not found: exchanging with peer copies\n"); J->use_rccl = false; }
		else if (!g_cc.busy && g_cc.devs == J->sdev && !g_cc.comm.empty()) { J->comm = g_cc.comm; J->comm_cached = g_cc.busy = true; }
		else {
			const ncclResult_t r = J->R.CommInitAll(J->comm.data(), S, J->sdev.data());
			if (r != ncclSuccess) { fprintf(stderr, "[W::yak_count] ncclCommInitAll: %s; exchanging with peer copies\n", J->R.GetErrorString ? J->R.GetErrorString(r) : "error"); J->use_rccl = false; }
			else if (!g_cc.busy) {       /* these become the process's (those of another device list go) */
				for (ncclComm_t c_ : g_cc.comm) if (c_) J->R.CommDestroy(c_);
				g_cc.comm = J->comm; g_cc.devs = J->sdev; J->comm_cached = g_cc.busy = true;
			}
		}
	}
	for (int s = 0; s < S; ++s) {
		if (hipSetDevice(J->sdev[s]) != hipSuccess || hipStreamCreate(&J->st[s]) != hipSuccess || hipStreamCreateWithFlags(&J->cp[s], hipStreamNonBlocking) != hipSuccess) return false;
		if (!J->use_rccl) for (int q = 0; q < S; ++q) if (q != s) (void)hipDeviceEnablePeerAccess(J->sdev[q], 0);
		for (int x = 0; x < 2; ++x) {
			/* (from the engine's pool: a process that counts again -- a benchmark's steps, the second pass of the filtered protocol -- finds these buffers there
			 * instead of asking the driver while the pool holds most of the device) */
			J->d_base[x][s] = J->ext_base ? 0 : (uint8_t*)yk_pool_get((size_t)J->chunk + 4096);
			J->d_send[x][s] = (uint64_t*)yk_pool_get((size_t)J->send_words * 8);
			J->d_recv[x][s] = J->recv_words ? (uint64_t*)yk_pool_get((size_t)J->recv_words * 8) : 0;
			if ((!J->ext_base && !J->d_base[x][s]) || !J->d_send[x][s] || (J->recv_words && !J->d_recv[x][s])) return false;
		}
	}
	(void)hipGetLastError();
	return true;
}

static void multi_close(MultiJob *J)
{
	for (int s = 0; s < J->S; ++s) {
		hipSetDevice(J->sdev[s]);
		for (int x = 0; x < 2; ++x) { if (!J->ext_base) yk_pool_release(J->d_base[x][s]); yk_pool_release(J->d_send[x][s]); yk_pool_release(J->d_recv[x][s]); J->d_base[x][s] = 0; J->d_send[x][s] = 0; J->d_recv[x][s] = 0; }
		if (J->st[s]) { hipStreamDestroy(J->st[s]); J->st[s] = 0; }
		if (J->cp[s]) { hipStreamDestroy(J->cp[s]); J->cp[s] = 0; }
		if (!J->comm_cached && s < (int)J->comm.size() && J->comm[s]) { J->R.CommDestroy(J->comm[s]); J->comm[s] = 0; }
	}
	if (J->comm_cached) { std::lock_guard<std::mutex> lk(g_cc.mu); g_cc.busy = false; J->comm_cached = false; }
}

/* One round on buffer set x: chunk s (fill[s] bytes, stream offset t0[s]) sits in slot s.  Three stages -- partition, exchange, feed -- that a caller
 * may run one after the other (multi_round) or overlapped round against round (yakamd_count_multi_dev: the partition of round b + 1 runs while round b
 * is on the wire and fed; the two buffer sets make that safe: set x is not touched again before the feed of the round that used it has returned). */
struct RoundPlan {
	bool tagged; int W;                                         /* 8-byte tagged records; words per record */
	std::vector<int64_t> fill; std::vector<uint64_t> t0;
	std::vector<std::vector<uint64_t> > bst;                    /* [slot][P + 1]: where the records of prefix p start in the slot's send buffer */
	std::vector<std::vector<uint64_t> > roff;                   /* [rank][slot]: where owner d's slice of chunk s lies in its slot's receive buffer (records) */
	std::string why; std::mutex why_mu;
	void note() { std::lock_guard<std::mutex> lk(why_mu); if (why.empty()) why = yakamd_last_error(); }   /* called on the thread that failed (the text is per thread) */
};

/* 8-byte tagged records: half the exchange; every owner must still be on the exclusive-ownership path */
static bool round_tagged(MultiJob *J, yak_ch_ext *e, int k, int pre, int create_new)
{
	bool tagged = create_new && yakamd_tagged_ok(k, pre) && !yk_knob("YAKAMD_MGPU_REC16", 0);
	for (int r = 0; r < J->N && tagged; ++r) tagged = e->sub[r] && yakamd_pass_fast(e->sub[r]);
	return tagged;
}

/* every slot groups the k-mers of its chunk by prefix (its own stream: nothing here waits for, or holds up, the owners' streams) */
static bool round_partition(MultiJob *J, int x, int k, int pre, int create_new, RoundPlan *R)
{
	const int P = J->P, S = J->S;
	R->W = create_new && !R->tagged ? 2 : 1;                    /* {hash, position}, or one word (tagged record / bare hash) */
	for (int s = 0; s < S; ++s) if (R->fill[s] * R->W > J->send_words) { fprintf(stderr, "[E::yak_count] a chunk of %lld positions does not fit the send buffer (%lld words): 16-byte records were not planned for\n", (long long)R->fill[s], (long long)J->send_words); return false; }
	R->bst.assign(S, std::vector<uint64_t>(P + 1, 0));
	std::vector<char> ok(S, 1);
	std::vector<std::thread> th;
	for (int s = 0; s < S; ++s) th.emplace_back([&, s]() {
		if (R->fill[s] <= 0) return;
		hipSetDevice(J->sdev[s]);
		const int64_t n = R->tagged ? yakamd_partition_tagged_dev(k, pre, J->d_base[x][s], R->fill[s], J->d_send[x][s], R->bst[s].data())
		                : create_new ? yakamd_partition_dev(k, pre, J->d_base[x][s], R->fill[s], J->d_send[x][s], R->bst[s].data())
		                             : yakamd_partition_hashes_dev(k, pre, J->d_base[x][s], R->fill[s], J->d_send[x][s], R->bst[s].data());
		if (n < 0) { ok[s] = 0; R->note(); }
	});
	for (auto &t : th) t.join();
	for (int s = 0; s < S; ++s) if (!ok[s]) return false;
	return true;
}

/* the slices of every chunk go to the slots of the ranks that own their prefixes.  Receive layout of a slot: for each rank it hosts (rank order), the
 * slices of the other slots' chunks (slot order) */
static bool round_exchange(MultiJob *J, int x, RoundPlan *R)
{
	const int N = J->N, P = J->P, S = J->S, W = R->W;
	const std::vector<std::vector<uint64_t> > &bst = R->bst;
	R->roff.assign(N, std::vector<uint64_t>(S, 0));
	std::vector<uint64_t> used(S, 0);
	for (int d = 0; d < N; ++d) {
		const int lo = d * (P / N), hi = (d + 1) * (P / N), sd = J->slot_of[d];
		for (int s = 0; s < S; ++s) { if (s == sd) continue; R->roff[d][s] = used[sd]; used[sd] += bst[s][hi] - bst[s][lo]; }
	}
	for (int s = 0; s < S; ++s) if ((int64_t)(used[s] * W) > J->recv_words) { fprintf(stderr, "[E::yak_count] device %d would receive %llu records in one round: prefixes too unevenly filled for YAKAMD_MGPU_CHUNK\n", J->sdev[s], (unsigned long long)used[s]); return false; }
	if (S <= 1) return true;
	bool ok = true;
	auto peer_copies = [&]() {
		for (int s = 0; s < S; ++s)
			for (int d = 0; d < N; ++d) {
				const int lo = d * (P / N), hi = (d + 1) * (P / N), sd = J->slot_of[d];
				const uint64_t cnt = (bst[s][hi] - bst[s][lo]) * W;
				if (cnt == 0 || s == sd) continue;
				hipSetDevice(J->sdev[sd]);
				if (hipMemcpyPeerAsync(J->d_recv[x][sd] + R->roff[d][s] * W, J->sdev[sd], J->d_send[x][s] + bst[s][lo] * W, J->sdev[s], cnt * 8, J->st[sd]) != hipSuccess) ok = false;
			}
	};
	auto wait_streams = [&]() { for (int s = 0; s < S; ++s) { hipSetDevice(J->sdev[s]); if (hipStreamSynchronize(J->st[s]) != hipSuccess) ok = false; } };
	if (J->use_rccl) {
		J->R.GroupStart();
		for (int s = 0; s < S; ++s)
			for (int d = 0; d < N; ++d) {
				const int lo = d * (P / N), hi = (d + 1) * (P / N), sd = J->slot_of[d];
				const uint64_t cnt = (bst[s][hi] - bst[s][lo]) * W;
				if (cnt == 0 || s == sd) continue;
				/* (the current device matches the communicator of every call, as the library's own examples do it) */
				hipSetDevice(J->sdev[s]);
				if (J->R.Send(J->d_send[x][s] + bst[s][lo] * W, cnt, ncclUint64, sd, J->comm[s], J->st[s]) != ncclSuccess) ok = false;
				hipSetDevice(J->sdev[sd]);
				if (J->R.Recv(J->d_recv[x][sd] + R->roff[d][s] * W, cnt, ncclUint64, s, J->comm[sd], J->st[sd]) != ncclSuccess) ok = false;
			}
		if (J->R.GroupEnd() != ncclSuccess) ok = false;
		wait_streams();
		if (!ok) {
			/* the collective library let the round down: the same slices as plain peer copies, from here on */
			fprintf(stderr, "[W::yak_count] RCCL exchange failed (%s): peer copies from now on\n", hipGetErrorString(hipGetLastError()));
			J->use_rccl = false; ok = true;
			/* nobody is handed these communicators again, and nobody destroys them (a communicator with a failed operation in it may hang in its destructor) */
			if (J->comm_cached) { std::lock_guard<std::mutex> lk(g_cc.mu); g_cc.comm.clear(); g_cc.devs.clear(); g_cc.busy = false; J->comm_cached = false; }
			J->comm.assign(S, 0);
			for (int s = 0; s < S; ++s) for (int q = 0; q < S; ++q) if (J->sdev[q] != J->sdev[s]) { hipSetDevice(J->sdev[s]); (void)hipDeviceEnablePeerAccess(J->sdev[q], 0); }
			(void)hipGetLastError();
			peer_copies();
			wait_streams();
		}
	} else {
		peer_copies();
		wait_streams();
	}
	if (!ok) fprintf(stderr, "[E::yak_count] exchange between the GPUs failed\n");
	return ok;
}

/* every owner takes its slices, in chunk order = stream order; owners that share a slot take turns (a feed may count a whole slice of the pass) */
static bool round_feed(MultiJob *J, int x, yak_ch_ext *e, int create_new, RoundPlan *R)
{
	const int N = J->N, P = J->P, S = J->S, W = R->W;
	const std::vector<std::vector<uint64_t> > &bst = R->bst;
	std::vector<char> ok(N, 1);
	std::vector<std::thread> th;
	for (int sd = 0; sd < S; ++sd) th.emplace_back([&, sd]() { for (int d = 0; d < N; ++d) if (J->slot_of[d] == sd) {
		hipSetDevice(J->dev[d]);
		const int lo = d * (P / N), hi = (d + 1) * (P / N);
		std::vector<uint64_t> ob(P + 1);
		for (int s = 0; s < S; ++s) {
			const uint64_t cnt = bst[s][hi] - bst[s][lo];
			if (cnt == 0) continue;
			for (int p = 0; p <= P; ++p) { const int q = p < lo ? lo : p > hi ? hi : p; ob[p] = bst[s][q] - bst[s][lo]; }
			const uint64_t *rec = s == sd ? J->d_send[x][s] + bst[s][lo] * W : J->d_recv[x][sd] + R->roff[d][s] * W;   /* the slice of the slot's own chunk is fed where the partition left it */
			const int rc = R->tagged ? yakamd_feed_partitioned_tagged_dev(e->sub[d], rec, (int64_t)cnt, ob.data(), R->t0[s], (uint64_t)R->fill[s], 0)
			             : create_new ? yakamd_feed_partitioned_dev(e->sub[d], rec, (int64_t)cnt, ob.data(), R->t0[s], (uint64_t)R->fill[s])
			                          : yakamd_count_partitioned_dev(e->sub[d], rec, (int64_t)cnt, ob.data());
			if (rc != 0) { ok[d] = 0; R->note(); }
		}
		if (hipStreamSynchronize(yk_ctx_stream(((yak_ch_ext*)e->sub[d])->ctx)) != hipSuccess) ok[d] = 0;   /* the copies out of this set's buffers are done before the set is filled again */
	} });
	for (auto &t : th) t.join();
	for (int r = 0; r < N; ++r) if (!ok[r]) return false;
	return true;
}

static bool multi_round(MultiJob *J, int x, yak_ch_ext *e, int k, int pre, int create_new, const std::vector<int64_t> &fill, const std::vector<uint64_t> &t0, std::string *why)
{
	RoundPlan R;
	R.fill = fill; R.t0 = t0;
	R.tagged = round_tagged(J, e, k, pre, create_new);
	const bool ok = round_partition(J, x, k, pre, create_new, &R) && round_exchange(J, x, &R) && round_feed(J, x, e, create_new, &R);
	if (!ok && why && why->empty()) *why = R.why;
	return ok;
}

static yak_ch_t *multi_table_new(const yak_copt_t *opt, int N, const std::vector<int> &dev);
yak_ch_t *yak_count_multi(const char *fn, const yak_copt_t *opt, yak_ch_t *h0, int N, const std::vector<int> &dev)
{
	FxReader fx;
	if (!fx.open_file(fn)) return 0;
	const int P = 1 << opt->pre;
	yak_ch_t *h = h0;
	const int create_new = h0 ? 0 : 1;
	if (h0 == 0) {                                             /* N tables, one per rank, each owning its prefix range */
		h = multi_table_new(opt, N, dev);
		if (!h) { fx.close_file(); return 0; }
	}
	yak_ch_ext *e = (yak_ch_ext*)h;
	MultiJob J;
	bool ok = multi_open(&J, N, P, dev);
	const int S = J.S;
	const double t_job0 = yk_realtime();
	for (int r = 0; r < N && ok; ++r) ok = yakamd_pass_begin(e->sub[r], create_new) == 0;
	/* the reader fills the chunks of set `cur` while a worker thread partitions, exchanges and feeds the set before it */
	std::vector<int64_t> fill[2] = { std::vector<int64_t>(S, 0), std::vector<int64_t>(S, 0) };
	std::vector<uint64_t> t0[2] = { std::vector<uint64_t>(S, 0), std::vector<uint64_t>(S, 0) };
	std::thread worker;
	bool worker_ok = true;
	std::string worker_why;                                    /* yakamd_last_error() is per thread: the round's text comes back with it */
	int cur = 0;
	uint64_t t_stream = 0;
	int64_t n_seq_tot = 0;
	int g = 0;                                                 /* the slot whose chunk is being filled */
	auto wait_worker = [&]() { if (worker.joinable()) worker.join(); if (!worker_ok) ok = false; };
	double t_sink = 0, t_round_wait = 0;                        /* YAKAMD_VERBOSE: where the reader's time goes */
	/* host -> device through two pinned staging buffers: while one is on its way over the bus the reader copies the next piece into the other (a
	 * copy from pageable memory is staged by the runtime anyway, but behind a synchronise per piece) */
	const size_t STG = (size_t)32 << 20;
	uint8_t *stg[2] = { 0, 0 };
	/* an event belongs to the device that was current when it was made and can only be recorded on a stream of that device: one per staging
	 * buffer AND slot, made with the slot's device current; stg_on[i] = the slot whose copy stream holds buffer i's last copy (-1: idle) */
	std::vector<hipEvent_t> stg_ev[2];
	int stg_on[2] = { -1, -1 };
	int stg_i = 0;
	for (int i = 0; i < 2 && ok; ++i) {
		ok = hipHostMalloc((void**)&stg[i], STG) == hipSuccess;
		stg_ev[i].assign(S, (hipEvent_t)0);
		for (int s = 0; s < S && ok; ++s) { hipSetDevice(J.sdev[s]); ok = hipEventCreateWithFlags(&stg_ev[i][s], hipEventDisableTiming) == hipSuccess; }
	}
	auto to_device = [&](int gdev, uint8_t *dst, const char *src, size_t n) -> bool {
		hipSetDevice(J.sdev[gdev]);
		for (size_t o = 0; o < n; o += STG) {
			const size_t m = std::min(STG, n - o);
			if (stg_on[stg_i] >= 0 && hipEventSynchronize(stg_ev[stg_i][stg_on[stg_i]]) != hipSuccess) return false;
			memcpy(stg[stg_i], src + o, m);
			if (hipMemcpyAsync(dst + o, stg[stg_i], m, hipMemcpyHostToDevice, J.cp[gdev]) != hipSuccess || hipEventRecord(stg_ev[stg_i][gdev], J.cp[gdev]) != hipSuccess) return false;
			stg_on[stg_i] = gdev; stg_i ^= 1;
		}
		return true;
	};
	auto copies_done = [&]() { for (int s = 0; s < S && ok; ++s) { hipSetDevice(J.sdev[s]); ok = hipStreamSynchronize(J.cp[s]) == hipSuccess; } };
	auto round = [&]() {
		const double tw0 = yk_realtime();
		copies_done();                                          /* the chunks of this set are on their devices */
		wait_worker();                                          /* at most one round in flight: its set becomes the one to fill next */
		t_round_wait += yk_realtime() - tw0;
		if (ok) {
			const int x = cur;
			worker = std::thread([&, x]() { std::string why; worker_ok = multi_round(&J, x, e, opt->k, opt->pre, create_new, fill[x], t0[x], &why); if (!worker_ok) worker_why = why; });
		}
		cur ^= 1;
		std::fill(fill[cur].begin(), fill[cur].end(), 0); g = 0;
	};
	/* a piece (whole sequences, each followed by '\n') goes to the chunk being filled; a chunk is closed between two
	 * sequences, or inside one that is longer than a whole chunk */
	auto take_piece_body = [&](const char *img, size_t n, int64_t ns) -> bool {
		n_seq_tot += ns;
		while (n > 0 && ok) {
			const size_t room = (size_t)(J.chunk - fill[cur][g]);
			size_t m = n, back = 0;
			if (n > room) {
				const void *nl = room ? memrchr(img, '\n', room) : 0;
				if (nl) m = (size_t)((const char*)nl - img) + 1;
				else if (fill[cur][g] > 0) { if (++g == S) round(); continue; }
				else {
					/* one sequence longer than a whole chunk (a chromosome beyond YAKAMD_MGPU_CHUNK bases): the chunk ends inside it and the
					 * next one starts k - 1 bases earlier -- the k-mers that end in this chunk are counted here, those that end behind it
					 * there (a chunk's first k - 1 positions complete no k-mer), and stream positions simply continue */
					m = room; back = (size_t)opt->k - 1;
				}
			}
			if (fill[cur][g] == 0) t0[cur][g] = t_stream;
			ok = to_device(g, J.d_base[cur][g] + fill[cur][g], img, m);
			fill[cur][g] += (int64_t)m;
			t_stream += m - back; img += m - back; n -= m - back;
			if (fill[cur][g] == J.chunk || n > 0) { if (++g == S) round(); }
		}
		return ok;
	};
	auto take_piece = [&](const char *img, size_t n, int64_t ns, const WinPack*) -> bool { const double t0 = yk_realtime(); const bool r = take_piece_body(img, n, ns); t_sink += yk_realtime() - t0; return r; };
	const int n_thr = parse_threads(opt->n_thread);
	ByteSource psrc; int psrc_fd = -1;
	bool par = parallel_source(fn, fx, n_thr, 1 << 20, &psrc, &psrc_fd);
	pgz::Reader gz;
	if (ok && par) ok = parse_parallel(&psrc, opt->k, n_thr, take_piece) && ok;
	else if (ok && gz_source(fn, fx, n_thr, &gz)) { par = true; ok = parse_gz(&gz, opt->k, n_thr, take_piece) && ok; }
	else if (ok) {
		std::vector<char> piece;
		int64_t l, ns = 0;
		for (;;) {
			if ((l = fx.fast(piece, opt->k)) == FxReader::NOT_FAST) {
				if ((l = fx.next()) < 0) break;
				if (l >= opt->k) { piece.insert(piece.end(), fx.seq.begin(), fx.seq.end()); piece.push_back('\n'); }
			}
			if (l >= opt->k) ++ns;
			if (piece.size() >= ((size_t)1 << 24)) { if (!take_piece(piece.data(), piece.size(), ns, 0)) break; piece.clear(); ns = 0; }
		}
		if (ok && !piece.empty()) take_piece(piece.data(), piece.size(), ns, 0);
	}
	if (ok) { bool any = false; for (int s = 0; s < S; ++s) any = any || fill[cur][s] > 0; if (any) round(); }
	wait_worker();
	const double t_fed = yk_realtime() - t_job0;
	for (int i = 0; i < 2; ++i) {
		if (stg_on[i] >= 0) (void)hipEventSynchronize(stg_ev[i][stg_on[i]]);
		for (hipEvent_t e_ : stg_ev[i]) if (e_) (void)hipEventDestroy(e_);
		if (stg[i]) (void)hipHostFree(stg[i]);
	}
	multi_close(&J);                                           /* the chunk and exchange buffers go before the passes finish: memory is tightest there */
	{	/* every rank finishes its pass: partitions, counting, layout -- side by side; ranks that share a device take turns, so that the
		 * scratch of only one of them is alive at a time (one device posing as N = the pass in N sweeps over prefix ranges: what lets a
		 * 5 Gb assembly through 288 GB) */
		std::vector<int64_t> n_ins(N, 0);
		std::vector<std::thread> th;
		std::vector<std::string> why(N);                       /* the error text is per thread: bring it back */
		for (int sd = 0; sd < S; ++sd) th.emplace_back([&, sd]() { for (int r = 0; r < N; ++r) if (J.slot_of[r] == sd) { n_ins[r] = yakamd_pass_end(e->sub[r]); if (n_ins[r] < 0) why[r] = yakamd_last_error(); } });
		for (auto &t : th) t.join();
		for (int r = 0; r < N; ++r) if (n_ins[r] < 0) fprintf(stderr, "[E::yak_count] rank %d of %d (device %d): %s\n", r, N, dev[r], why[r].c_str());
		for (int r = 0; r < N; ++r) { if (n_ins[r] < 0) ok = false; else e->sub[r]->tot += (uint64_t)n_ins[r]; }
	}
	multi_tot(h);
	if (getenv("YAKAMD_VERBOSE") && atoi(getenv("YAKAMD_VERBOSE")) > 0) {
		fprintf(stderr, "[yak_amd] %d ranks: input read, dealt and fed by %.3f s (%d parser threads; %.3f s inside the sink that copies the pieces to the devices, %.3f s of it waiting for copies and the round before), the ranks' passes finished by %.3f s\n",
		        N, t_fed, n_thr, t_sink, t_round_wait, yk_realtime() - t_job0);
		for (int s = 0; s < S; ++s) { hipSetDevice(J.sdev[s]); yk_pool_report("the job"); }
	}
	fprintf(stderr, "[M::%s::%.3f*%.2f] %ld sequences in total; %ld distinct k-mers in the hash table (%d GPUs, %s)\n", "yak_count",
	        yk_realtime(), yk_cputime() / (yk_realtime() + 1e-9), (long)n_seq_tot, (long)h->tot, N, S == 1 ? "one device: nothing exchanged" : J.use_rccl ? (J.loopback ? "grouped send / recv served by the in-process test rig" : "RCCL exchange") : "peer copies");
	if (psrc_fd >= 0) ::close(psrc_fd);
	fx.close_file();
	if (!ok) { fprintf(stderr, "[E::yak_count] %s\n", !worker_why.empty() ? worker_why.c_str() : yakamd_last_error()); if (!h0) yak_ch_destroy(h); return 0; }
	return h;
}

/* a table sharded over N ranks (dev[r] = device of rank r), every rank owning its prefix range */
static yak_ch_t *multi_table_new(const yak_copt_t *opt, int N, const std::vector<int> &dev)
{
	const int P = 1 << opt->pre;
	yak_ch_ext *e = (yak_ch_ext*)calloc(1, sizeof(*e));
	e->magic = EXT_MAGIC; e->n_sub = N; e->sub = (yak_ch_t**)calloc(N, sizeof(yak_ch_t*));
	yak_ch_t *h = &e->pub;
	h->k = opt->k; h->pre = opt->pre;
	h->h = (yak_ch1_t*)calloc((size_t)P, sizeof(yak_ch1_t));
	bool ok = true;
	for (int r = 0; r < N && ok; ++r) {
		yk_ctx_next_device(dev[r]);
		e->sub[r] = yak_ch_init(opt->k, opt->pre, opt->bf_n_hash, opt->bf_shift);
		ok = e->sub[r] && yakamd_set_shard(e->sub[r], r * (P / N), (r + 1) * (P / N)) == 0;
	}
	if (!ok) { for (int r = 0; r < N; ++r) if (e->sub[r]) yak_ch_destroy(e->sub[r]); free(e->sub); free(h->h); free(e); return 0; }
	e->ctx = 0;                                               /* no context of its own: every yakamd_* entry point refuses a sharded table instead of working on one shard */
	h->n_hash = e->sub[0]->n_hash; h->n_shift = e->sub[0]->n_shift;
	for (int p = 0; p < P; ++p) h->h[p].b = e->sub[0]->h[p].b;   /* descriptors only: "has a filter" for callers that look */
	return h;
}

/* The same job with its input already on the devices (the benchmark's N-GPU mode; a caller with its own reader): the stream is cut into rounds of
 * one chunk per DEVICE -- chunk s of round b lies at d_chunk[b * S + s] on the s-th distinct device of `dev` (n_bytes[b * S + s] bytes of the base
 * image, at most 2^31 - 4096; 0 = none), and the stream order is round by round, device by device, exactly as yak_count() deals a file.  h0 == 0:
 * a new table sharded over the n_rank ranks (dev[r] = device of rank r; several ranks may share a device) comes back; h0 != 0: its k-mers are counted
 * (count.c:155-157).  exchange_out (may be 0): 1 = RCCL grouped send / recv, 2 = peer copies, 3 = the grouped calls served by the in-process test rig, 0 = one device, nothing exchanged.  The caller keeps
 * the chunks alive until the call returns */
extern "C" yak_ch_t *yakamd_count_multi_dev(const yak_copt_t *opt, yak_ch_t *h0, int n_rank, const int *dev_of_rank, int n_rounds,
                                            const void *const *d_chunk, const int64_t *n_bytes, int *exchange_out)
{
	const int P = 1 << opt->pre, N = n_rank;
	if (N < 1 || P % N) { fprintf(stderr, "[E::yakamd_count_multi_dev] %d ranks do not divide the %d sub-tables\n", N, P); return 0; }
	std::vector<int> dev(dev_of_rank, dev_of_rank + N);
	if (h0) {
		yak_ch_ext *e0 = (yak_ch_ext*)h0;
		if ((e0->n_sub > 1 ? e0->n_sub : 1) != N) { fprintf(stderr, "[E::yakamd_count_multi_dev] the table is sharded over %d ranks, not %d\n", e0->n_sub > 1 ? e0->n_sub : 1, N); return 0; }
		assert(h0->k == opt->k && h0->pre == opt->pre);
	}
	const int create_new = h0 ? 0 : 1;
	yak_ch_t *h = h0;
	if (!h && N == 1) { yk_ctx_next_device(dev[0]); h = yak_ch_init(opt->k, opt->pre, opt->bf_n_hash, opt->bf_shift); }   /* one rank: an ordinary table on that device (the same driver: rounds of chunks, partition, feed) */
	else if (!h) h = multi_table_new(opt, N, dev);
	if (!h) return 0;
	yak_ch_ext *e = (yak_ch_ext*)h, one_rank;
	yak_ch_t *only[1] = { h };
	if (N == 1) { memset(&one_rank, 0, sizeof(one_rank)); one_rank.n_sub = 1; one_rank.sub = only; e = &one_rank; }   /* the rounds address the owners as e->sub[rank] */
	MultiJob J;
	J.S = 0;
	int64_t cmax = 4096;
	{	/* the distinct devices, in rank order: that is the order of the chunks inside a round */
		std::vector<int> sd;
		for (int r = 0; r < N; ++r) if (std::find(sd.begin(), sd.end(), dev[r]) == sd.end()) sd.push_back(dev[r]);
		for (int i = 0; i < n_rounds * (int)sd.size(); ++i) cmax = std::max<int64_t>(cmax, n_bytes[i]);
	}
	if (cmax > ((int64_t)1 << 31) - 4096) { fprintf(stderr, "[E::yakamd_count_multi_dev] a chunk holds at most 2^31 - 4096 stream positions\n"); if (!h0) yak_ch_destroy(h); return 0; }
	const bool tagged_only = create_new && yakamd_tagged_ok(opt->k, opt->pre) && !yk_knob("YAKAMD_MGPU_REC16", 0) && env_fast_default();
	bool ok = multi_open(&J, N, P, dev, cmax, tagged_only || !create_new);
	const int S = J.S;
	yk_realtime();
	for (int r = 0; r < N && ok; ++r) ok = yakamd_pass_begin(e->sub[r], create_new) == 0;
	std::string why;
	uint64_t t_stream = 0;
	/* rounds overlapped: while round b is exchanged and fed (this thread), a second thread partitions round b + 1 into the other buffer set, on the
	 * slots' partition streams.  Set (b + 1) & 1 was last used by round b - 1, whose feed has returned; whether round b + 1 uses tagged records is
	 * settled before its partition starts, i.e. before the feed of round b can take an owner off the exclusive-ownership path -- such a pass fails at
	 * the feed of round b + 1 with the engine's own message, it does not mix formats.  YAKAMD_MGPU_NO_OVERLAP (test switch): one stage after the other. */
	const bool overlap = !yk_knob("YAKAMD_MGPU_NO_OVERLAP", 0);
	RoundPlan R[2];
	auto prepare = [&](int b) {
		RoundPlan &r = R[b & 1];
		r.fill.assign(S, 0); r.t0.assign(S, 0); r.why.clear();
		for (int s = 0; s < S; ++s) {
			r.fill[s] = n_bytes[(size_t)b * S + s];
			r.t0[s] = t_stream; t_stream += (uint64_t)r.fill[s];
			J.d_base[b & 1][s] = (uint8_t*)d_chunk[(size_t)b * S + s];
		}
		r.tagged = round_tagged(&J, e, opt->k, opt->pre, create_new);
	};
	std::thread part;
	bool part_ok = true;
	double t_part = 0, t_wait = 0, t_exch = 0, t_feed = 0;        /* YAKAMD_VERBOSE: the partition thread's time; this thread's wait for it, exchange, feed */
	auto start_partition = [&](int b) { part = std::thread([&, b]() { const double t0 = yk_realtime(); part_ok = round_partition(&J, b & 1, opt->k, opt->pre, create_new, &R[b & 1]); t_part += yk_realtime() - t0; }); };
	const double t_rounds0 = yk_realtime();
	if (ok && n_rounds > 0) { prepare(0); start_partition(0); }
	for (int b = 0; b < n_rounds && ok; ++b) {
		double t0 = yk_realtime();
		if (part.joinable()) part.join();
		ok = part_ok;
		if (ok && b + 1 < n_rounds) { prepare(b + 1); start_partition(b + 1); if (!overlap) { part.join(); ok = part_ok; } }
		t_wait += yk_realtime() - t0; t0 = yk_realtime();
		ok = ok && round_exchange(&J, b & 1, &R[b & 1]);
		t_exch += yk_realtime() - t0; t0 = yk_realtime();
		ok = ok && round_feed(&J, b & 1, e, create_new, &R[b & 1]);
		t_feed += yk_realtime() - t0;
		if (!ok && why.empty()) why = !R[b & 1].why.empty() ? R[b & 1].why : R[(b + 1) & 1].why;
	}
	if (part.joinable()) part.join();
	const double t_rounds = yk_realtime() - t_rounds0;
	const int exch = S == 1 ? 0 : J.use_rccl ? (J.loopback ? 3 : 1) : 2;
	for (int x = 0; x < 2; ++x) for (int s = 0; s < S; ++s) J.d_base[x][s] = 0;
	multi_close(&J);
	{
		std::vector<int64_t> n_ins(N, 0);
		std::vector<std::thread> th;
		std::vector<std::string> whyr(N);
		for (int sd = 0; sd < S; ++sd) th.emplace_back([&, sd]() { for (int r = 0; r < N; ++r) if (J.slot_of[r] == sd) { n_ins[r] = yakamd_pass_end(e->sub[r]); if (n_ins[r] < 0) whyr[r] = yakamd_last_error(); } });
		for (auto &t : th) t.join();
		for (int r = 0; r < N; ++r) if (n_ins[r] < 0) { fprintf(stderr, "[E::yakamd_count_multi_dev] rank %d of %d (device %d): %s\n", r, N, dev[r], whyr[r].c_str()); ok = false; }
		for (int r = 0; r < N; ++r) if (n_ins[r] >= 0) e->sub[r]->tot += (uint64_t)n_ins[r];
	}
	if (N > 1) multi_tot(h);
	if (yk_knob("YAKAMD_VERBOSE", 0) > 0) {
		fprintf(stderr, "[yak_amd] %d rounds in %.3f s (%s): partitions %.3f s on their thread; this thread waited %.3f s for them, exchanged for %.3f s, fed for %.3f s; the ranks' passes finished by %.3f s\n",
		        n_rounds, t_rounds, overlap ? "partition of round b + 1 under exchange and feed of round b" : "stage by stage", t_part, t_wait, t_exch, t_feed, yk_realtime() - t_rounds0);
		for (int s = 0; s < S; ++s) if (s == 0 || J.sdev[s] != J.sdev[s - 1]) { hipSetDevice(J.sdev[s]); yk_pool_report("the job"); }
	}
	if (exchange_out) *exchange_out = exch;
	fprintf(stderr, "[M::%s::%.3f*%.2f] %d rounds of device-resident chunks; %ld distinct k-mers in the hash table (%d ranks, %s)\n", "yakamd_count_multi_dev",
	        yk_realtime(), yk_cputime() / (yk_realtime() + 1e-9), n_rounds, (long)h->tot, N, exch == 0 ? "one device: nothing exchanged" : exch == 1 ? "RCCL exchange" : exch == 3 ? "grouped send / recv served by the in-process test rig" : "peer copies");
	if (!ok) { fprintf(stderr, "[E::yakamd_count_multi_dev] %s\n", !why.empty() ? why.c_str() : yakamd_last_error()); if (!h0) yak_ch_destroy(h); return 0; }
	return h;
}

/* reference count.c:147-166 */
