import sys, os, hashlib, struct
sys.path.insert(0, os.getcwd())
import yak_amd, __graft_entry__ as ge
L = yak_amd.lib()
for n_reads in (200000, 1000000, 3000000):
    img = ge._synth(n_reads, 150, 5 * n_reads, 42)
    nb = len(img)
    for bf in (0, 30):
        a, atot = yak_amd.count_protocol_host(img, k=31, bf_shift=bf)
        d = L.yakamd_dev_alloc(nb); L.yakamd_memcpy_h2d(d, img, nb)
        xh, xt = L.yakamd_dev_alloc(nb * 8), L.yakamd_dev_alloc(nb * 4)
        t = yak_amd.Table(31, 10, 4, bf)
        def one(create):
            assert L.yakamd_pass_begin(t.h, create) == 0
            n = L.yakamd_extract_dev(31, d, nb, xh, xt, 10, 0, 1024, None)
            assert L.yakamd_feed_hashed_dev(t.h, xh, xt, n, 0, nb) == 0
            r = L.yakamd_pass_end(t.h); assert r >= 0; t.h.contents.tot += r
            return n
        n = one(1)
        st = t.stats()
        if bf: t.destroy_bf(); t.clear(); one(0); t.shrink(2, 1023)
        b = t.dump_bytes(); btot = t.tot; t.close()
        for p in (d, xh, xt): L.yakamd_dev_free(p)
        print(n_reads, bf, "inst", n, "same" if a == b else "DIFF", atot, btot, {k: round(v, 1) for k, v in st.items() if k.startswith("ms_i") or k.startswith("n_")})
