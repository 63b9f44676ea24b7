#!/usr/bin/env python3
"""Can the exact-layout replay start ABOVE the smallest capacities?

khashl (reference khashl.h:152-221) grows a table 4 -> 8 -> 16 -> ... ; a put-call that finds count >= 0.75 capacity doubles first (khashl.h:202), and a
doubling re-inserts the old table's keys IN OLD SLOT ORDER with kick-outs (khashl.h:171-189).  Hypothesis H(C0): the layout right after the doubling
to capacity C0 equals the layout of inserting the same keys, first come first served in time order, into an EMPTY table of capacity C0 -- if it held, the
replay could place the first 0.375 C0 keys of a sub-table directly and only replay the doublings above C0.

This script is a literal Python model of khashl's put / resize on 32-bit hashes (yak_ch_hash = key >> 10, __kh_h2b = hash * 2654435769 >> (32 - bits))
and (1) prints the smallest counter-example it finds by exhaustive search over home slots, (2) measures over random key sets how often H(C0) fails and
how many slots differ, for C0 = 8 ... 8192.  Result (committed in profiles/r06_experiments.txt): H fails already for C0 = 8 with two keys, and for random
keys fails with probability -> 1 as C0 grows (the expected number of differing slots grows linearly with C0): every doubling matters to the bytes.  What
CAN be skipped is the cost, not the history: tables up to 8 Ki slots replay their whole history inside one workgroup's LDS (k_replay), so the streaming
stages (k_r2_double / place) only start at 16 Ki slots -- the replay already "starts above the smallest capacities" in the only sense that is exact.
"""
import random
import sys

M32 = 0xFFFFFFFF


def h2b(h, bits):
    return ((h * 2654435769) & M32) >> (32 - bits)


class Kh:
    """khashl set of 32-bit hashes, keys == hashes (what matters to the layout)"""
    def __init__(self):
        self.bits, self.count, self.used, self.keys = 0, 0, None, None

    def n_buckets(self):
        return (1 << self.bits) if self.keys is not None else 0

    def resize(self, new_n):
        j, x = 0, new_n
        while x > 1:
            x >>= 1; j += 1
        if new_n & (new_n - 1):
            j += 1
        nb = max(j, 2)
        new_n = 1 << nb
        if self.count > (new_n >> 1) + (new_n >> 2):
            return
        new_used = [False] * new_n
        n = self.n_buckets()
        if self.keys is None:
            self.keys, self.used = [None] * new_n, [False] * 0
        if n < new_n:
            self.keys = self.keys + [None] * (new_n - n)
        mask = new_n - 1
        for j in range(n):
            if not self.used[j]:
                continue
            key = self.keys[j]
            self.used[j] = False
            while True:
                i = h2b(key, nb)
                while new_used[i]:
                    i = (i + 1) & mask
                new_used[i] = True
                if i < n and self.used[i]:
                    self.keys[i], key = key, self.keys[i]
                    self.used[i] = False
                else:
                    self.keys[i] = key
                    break
        self.used, self.bits = new_used, nb

    def put(self, key):
        n = self.n_buckets()
        if self.count >= (n >> 1) + (n >> 2):
            self.resize(n + 1)
            n = self.n_buckets()
        mask = n - 1
        i = last = h2b(key, self.bits)
        while self.used[i] and self.keys[i] != key:
            i = (i + 1) & mask
            if i == last:
                break
        if not self.used[i]:
            self.keys[i], self.used[i] = key, True
            self.count += 1

    def layout(self):
        return [self.keys[i] if self.used[i] else None for i in range(self.n_buckets())]


def through_history(keys, upto_bits):
    """the keys put in order; the layout right after the doubling that reaches 1 << upto_bits (before the put-call that triggered it places its key)"""
    t = Kh()
    for idx, k in enumerate(keys):
        n = t.n_buckets()
        if t.count >= (n >> 1) + (n >> 2) and n * 2 == (1 << upto_bits) or (n == 0 and upto_bits == 2):
            t.resize(n + 1)
            return t.layout(), idx
        t.put(k)
    return None, len(keys)


def direct(keys, bits):
    t = Kh()
    t.keys, t.used, t.bits = [None] * (1 << bits), [False] * (1 << bits), bits
    mask = (1 << bits) - 1
    for k in keys:
        i = h2b(k, bits)
        while t.used[i]:
            i = (i + 1) & mask
        t.keys[i], t.used[i] = k, True
    return t.layout()


def smallest(bits, need_no_wrap, seed=1, tries=200000):
    """the first random key sequence whose layout right after the doubling to 1 << bits differs from the direct placement; need_no_wrap: no key's probe
    sequence (old or new table) may cross the end of the array, so the difference is not an artefact of the wrap-around"""
    rng = random.Random(seed)
    n_keys = (1 << bits)
    for _ in range(tries):
        keys = [rng.getrandbits(32) for _ in range(n_keys)]
        hist, n_in = through_history(keys, bits)
        dire = direct(keys[:n_in], bits)
        if hist == dire:
            continue
        if need_no_wrap:
            old = through_history(keys, bits - 1)[0] if bits > 3 else None
            if any(lay is not None and any(k is not None and h2b(k, b_) > i for i, k in enumerate(lay)) for lay, b_ in ((hist, bits), (dire, bits), (old, bits - 1))):
                continue
        return keys[:n_in], hist, dire
    return None


def survey(bits_list, trials, seed):
    rng = random.Random(seed)
    rows = []
    for bits in bits_list:
        fails, diff_slots = 0, 0
        n_keys = (1 << bits)                                  # more than enough put-calls to reach the doubling to 1 << bits
        for _ in range(trials):
            keys = [rng.getrandbits(32) for _ in range(n_keys)]
            hist, n_in = through_history(keys, bits)
            dire = direct(keys[:n_in], bits)
            d = sum(1 for x, y in zip(hist, dire) if x != y)
            fails += d > 0
            diff_slots += d
        rows.append((bits, trials, fails, diff_slots / trials))
    return rows


if __name__ == "__main__":
    for bits, nw in ((3, False), (4, True)):
        r = smallest(bits, nw)
        if r is None:
            print(f"no counter-example found at C0 = {1 << bits}" + (" without wrap-around" if nw else ""))
            continue
        keys, hist, dire = r
        print(f"counter-example, capacity {1 << (bits - 1)} -> {1 << bits}" + (" (no key of either layout sits below its home slot)" if nw else "") + ", hashes in put order:", ["%08x" % k for k in keys])
        print(f"  homes at {1 << (bits - 1)} slots:", [h2b(k, bits - 1) for k in keys], f" homes at {1 << bits} slots:", [h2b(k, bits) for k in keys])
        print("  through the doublings:", ["%08x" % k if k is not None else "-" for k in hist])
        print("  direct, time order   :", ["%08x" % k if k is not None else "-" for k in dire])
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    print("random 32-bit hashes: layout right after the doubling to C0 vs direct first-come-first-served placement of the same keys into an empty table of C0 slots")
    for bits, tr, fails, mean in survey([3, 4, 6, 8, 10, 12, 13], trials, 7):
        print(f"  C0 = {1 << bits:5d}: {fails:4d} of {tr} tables differ, {mean:8.2f} slots differ on average ({mean / (0.375 * (1 << bits)) * 100:5.1f} % of the keys)")
