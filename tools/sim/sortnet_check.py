"""0-1-principle check of the sorting networks and of the merge step used by the kNN drain (csrc/knn.hip: sort16, sort8,
bitonic_merge16): run on the CPU, no GPU needed.  Prints OK lines or raises."""
import itertools, random, re, os

src = open(os.path.join(os.path.dirname(__file__), "..", "..", "myria3d_amd", "csrc", "knn.hip")).read()

def table(name, n):
    body = src[src.index(f"void {name}("):]
    body = body[body.index("= {") + 3: body.index("};")]
    return [(int(a), int(b)) for a, b in re.findall(r"\{(\d+),\s*(\d+)\}", body)]

def sorts(net, n):
    for bits in range(1 << n):
        v = [(bits >> i) & 1 for i in range(n)]
        for a, b in net:
            if v[a] > v[b]: v[a], v[b] = v[b], v[a]
        if any(v[i] > v[i + 1] for i in range(n - 1)): return False
    return True

n16, n8 = table("sort16", 16), table("sort8", 8)
assert len(n16) == 60 and sorts(n16, 16); print("sort16: 60 compare-exchanges, sorts every 0-1 input")
assert len(n8) == 19 and sorts(n8, 8); print("sort8: 19 compare-exchanges, sorts every 0-1 input")

def bitonic_merge16(v):
    s = 8
    while s >= 1:
        for i in range(16):
            if (i & s) == 0 and v[i] > v[i + s]: v[i], v[i + s] = v[i + s], v[i]
        s >>= 1

rng = random.Random(0)
INF = float("inf")
for trial in range(20000):
    m = rng.randint(0, 16)
    nb = rng.randint(0, 16)
    best = sorted(rng.random() for _ in range(nb)) + [INF] * (16 - nb)
    q = sorted([rng.random() for _ in range(m)] + [INF] * (16 - m))
    ref = sorted(best + q)[:16]
    v = [min(best[i], q[15 - i]) for i in range(16)]
    bitonic_merge16(v)
    assert v == ref
    if m <= 8:  # 8-slot variant: best[0..7] untouched by the first half-cleaner
        q8 = q[:8]
        v = best[:8] + [min(best[i], q8[15 - i]) for i in range(8, 16)]
        bitonic_merge16(v)
        assert v == ref
print("merge: 16 + 16 and 16 + 8 keys -> the 16 smallest, ascending (20000 random cases incl. +inf padding)")
