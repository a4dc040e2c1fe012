"""Register / LDS / spill figures of the kernels in libgnnvc_hip.so (the gfx950 code object's notes).
python tools/kernel_resources.py [name-substring ...]"""
import pathlib, struct, subprocess, sys, tempfile, re
ROOT = pathlib.Path(__file__).resolve().parents[1]
data = (ROOT / "gnn-mwvc_amd" / "libgnnvc_hip.so").read_bytes()
i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", data, i + 24)[0]
off = i + 32
co = None
for _ in range(n):
    o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
    trip = data[off:off + tl].decode(); off += tl
    if "gfx950" in trip:
        co = data[i + o:i + o + sz]
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(co); f.flush()
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
want = sys.argv[1:]
for block in notes.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", block).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if want and not any(w in dem for w in want):
        continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", block).group(1)
    print(f"{dem[:110]}: vgpr {g('vgpr_count')} sgpr {g('sgpr_count')} lds {g('group_segment_fixed_size')} scratch {g('private_segment_fixed_size')} "
          f"sgpr_spill {g('sgpr_spill_count')} vgpr_spill {g('vgpr_spill_count')}")
