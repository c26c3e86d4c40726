"""Does the apply's speed depend on HOW the cell arrays were allocated?  bench.py (host layer: one hipMalloc of exactly
n*8 bytes per array) measured 12.8 us on a box where tools/apply_shape_sweep.py (torch tensors) measured 9.3 us for the same
kernel in the same session.  Rings of 9 (src, dst) pairs allocated in different ways, K applies / copies between events."""
import sys, pathlib, ctypes as C
import torch
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from hyteg_amd import capi, host

L = 8
capi.lib(); capi.prepare_level(L)
lib = capi.lib()
n = capi.cell_size(L)
nbuf, K, R = 9, 500, 5
st = torch.cuda.current_stream().cuda_stream
w = [0.1 * (k + 1) for k in range(15)]; w[7] = -3.0
E0, E1 = capi.event_create_timing(), capi.event_create_timing()
MB2 = 2 << 20

def hmalloc(nbytes):
    p = C.c_void_p()
    assert lib.hyteg_hip_malloc(C.byref(p), nbytes) == 0
    return p.value

def fill(ptr, nbytes):
    # random contents via a torch tensor copy
    t = torch.rand(n, dtype=torch.float64, device="cuda")
    assert lib.hyteg_hip_copy(C.c_void_p(ptr), C.c_void_p(t.data_ptr()), n * 8, C.c_void_p(st)) == 0
    torch.cuda.synchronize()

def ring_exact():
    return [(hmalloc(n * 8), hmalloc(n * 8)) for _ in range(nbuf)], None
def ring_exact_interleaved_all_src_first():
    s = [hmalloc(n * 8) for _ in range(nbuf)]; d = [hmalloc(n * 8) for _ in range(nbuf)]
    return list(zip(d, s)), None
def ring_rounded():
    r = -(-n * 8 // MB2) * MB2
    return [(hmalloc(r), hmalloc(r)) for _ in range(nbuf)], None
def ring_arena(pad, m=None):
    m = m or nbuf
    r = -(-n * 8 // MB2) * MB2 + pad
    base = hmalloc(2 * m * r + MB2)
    b = -(-base // MB2) * MB2
    return [(b + (2 * k) * r, b + (2 * k + 1) * r) for k in range(m)], None
def ring_exact_m(m):
    return [(hmalloc(n * 8), hmalloc(n * 8)) for _ in range(m)], None
def ring_torch(m=None):
    m = m or nbuf
    keep = [(torch.empty(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda")) for _ in range(m)]
    return [(a.data_ptr(), b.data_ptr()) for a, b in keep], keep

_storage = None
def ring_host():
    global _storage
    if _storage is None:
        host.lib()
        _storage = host.Storage.from_gmsh(ROOT / "hyteg_amd/data/meshes/tet_1el.msh", 0, 1)
        _storage.set_stream(st)
    s = [host.P1Function(_storage, f"s{k}", L, L) for k in range(nbuf)]
    d = [host.P1Function(_storage, f"d{k}", L, L) for k in range(nbuf)]
    return [(d[k].cell_pointer(0, L), s[k].cell_pointer(0, L)) for k in range(nbuf)], (s, d)

def timeit(pairs, what):
    ring = capi.calib_copy_ring(pairs, n, True, st)
    def app(first, count, e0, e1):
        capi.event_record(e0, st)
        for k in range(first, first + count):
            d, s = pairs[k % len(pairs)]
            capi.p1_apply_cell(d, s, L, w, 0, st)
        capi.event_record(e1, st)
    fn = app if what == "apply" else ring
    fn(0, 2 * len(pairs), E0, E1); torch.cuda.synchronize()
    out = []
    for _ in range(R):
        torch.cuda.synchronize()
        fn(0, K, E0, E1)
        out.append(capi.event_elapsed_ms(E0, E1) * 1e3 / K)
    return sorted(out)[R // 2]

for name, mk in [("host layer: P1Function arrays (bench.py's ring)", ring_host),
                 ("hipMalloc(n*8) per array, dst/src alternating (host layer)", ring_exact),
                 ("hipMalloc(n*8), all src first then all dst", ring_exact_interleaved_all_src_first),
                 ("hipMalloc(rounded up to 2 MiB)", ring_rounded),
                 ("one arena, arrays at 2 MiB-aligned offsets", lambda: ring_arena(0)),
                 ("one arena, 2 MiB-aligned + 64 KiB stagger per array", lambda: ring_arena(65536)),
                 ("one arena, 2 MiB-aligned + 4 KiB + 256 B stagger", lambda: ring_arena(4096 + 256)),
                 ("torch.empty", ring_torch),
                 ("torch.empty, ring of 6 pairs (275 MB: the shape sweep's)", lambda: ring_torch(6)),
                 ("torch.empty, ring of 3 pairs (137 MB < Infinity Cache)", lambda: ring_torch(3)),
                 ("torch.empty, ring of 18 pairs (824 MB)", lambda: ring_torch(18)),
                 ("hipMalloc per array, ring of 18 pairs", lambda: ring_exact_m(18)),
                 ("ONE arena of 18 pairs (864 MB in one hipMalloc)", lambda: ring_arena(0, 18)),
                 ("ONE arena of 36 pairs (1.7 GB in one hipMalloc)", lambda: ring_arena(0, 36)),
                 ("torch.empty, ring of 12 pairs (550 MB)", lambda: ring_torch(12)),
                 ("torch.empty, ring of 14 pairs (641 MB)", lambda: ring_torch(14)),
                 ("hipMalloc(n*8) per array again", ring_exact),
                 ("host layer: P1Function arrays again", ring_host)]:
    pairs, keep = mk()
    for d, s in pairs:
        fill(s, n * 8)
    a, c = timeit(pairs, "apply"), timeit(pairs, "copy")
    a2 = timeit(pairs, "apply")
    offs = sorted({(d % MB2) // 4096 for d, s in pairs} | {(s % MB2) // 4096 for d, s in pairs})
    print(f"{name:62s} apply {a:7.3f} / {a2:7.3f} us  copy {c:7.3f} us   VA mod 2MiB (4K pages): {offs[:6]}{'...' if len(offs) > 6 else ''}"
          f"  first pair dst-src = {pairs[0][0] - pairs[0][1]:+d}", flush=True)
