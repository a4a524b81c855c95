"""Register / scratch / LDS table of every kernel in megaportrait-hack_amd/csrc/*.hip at HEAD: compiles each
translation unit for gfx950 to assembly (no GPU needed) and reads the amdhsa kernel metadata hipcc emits.
usage: python tools/register_table.py [out.json]     (default: print)
The CPU test tests/test_host.py::test_hot_kernels_have_no_scratch uses collect() on the hot translation units."""
import concurrent.futures, hashlib, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "megaportrait-hack_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-S", "--cuda-device-only", "-o", "-"]
EXTRA = {"conv3d_f16x3_wino_pp.hip": ["-fno-slp-vectorize"], "conv3d_f16x3_wino_bt.hip": ["-fno-slp-vectorize"]}   # (mirrors csrc/build.sh)
KEYS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
        ".group_segment_fixed_size", ".max_flat_workgroup_size")


def demangle(names):
    """kernel name + integer template arguments, e.g. conv3d_k3_f16x3_kernel<4, 8, 16, 8, 1> (binutils' c++filt does not know the _Float16
    mangling DF16_ in the argument lists, so the names are decoded here: _ZN5mphip<len><name>[I<Li..E / Lb..E>...E]...)."""
    out = []
    for n in names:
        m = re.match(r"_ZN5mphip(\d+)", n)
        if not m:
            out.append(n)
            continue
        ln = int(m.group(1))
        base, rest = n[m.end():m.end() + ln], n[m.end() + ln:]
        args = ""
        if rest.startswith("I"):
            vals = re.findall(r"L[ib](\d+)E", rest[:rest.index("EE") + 1] if "EE" in rest else rest)
            args = "<" + ", ".join(vals) + ">"
        out.append(base + args)
    return out


def one(path):
    asm = subprocess.run([HIPCC] + FLAGS + EXTRA.get(os.path.basename(path), []) + [path], capture_output=True, text=True)
    if asm.returncode != 0:
        raise RuntimeError(f"{path}: {asm.stderr[-2000:]}")
    kernels, cur = [], None
    in_meta = False
    for line in asm.stdout.splitlines():
        if line.startswith("amdhsa.kernels:"):
            in_meta = True
            continue
        if not in_meta:
            continue
        m = re.match(r"\s+(- )?(\.[a-z_]+):\s*(.*)$", line)
        if not m:
            if line.startswith("amdhsa.") and not line.startswith("amdhsa.kernels"):
                in_meta = False
            continue
        key, val = m.group(2), m.group(3).strip()
        if key == ".name":
            cur = {"name": val}
            kernels.append(cur)
        elif cur is not None and key in KEYS:
            cur[key[1:]] = int(val)
    # .args lists come before .name inside an entry: fields seen before the first .name of an entry belong to it -> re-parse by entry
    entries = re.split(r"\n  - ", asm.stdout[asm.stdout.find("amdhsa.kernels:"):])[1:]
    kernels = []
    for e in entries:
        name = re.search(r"\.name:\s*(\S+)", e)
        if not name:
            continue
        k = {"name": name.group(1)}
        for key in KEYS:
            mm = re.search(re.escape(key) + r":\s*(\d+)", e)
            if mm:
                k[key[1:]] = int(mm.group(1))
        kernels.append(k)
    for k, d in zip(kernels, demangle([k["name"] for k in kernels])):
        k["demangled"] = d
    return {"sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(), "kernels": kernels, "asm": asm.stdout if os.environ.get("MPHIP_KEEP_ASM") else None}


def collect(files=None):
    files = files or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(files))) as ex:
        res = list(ex.map(lambda f: one(os.path.join(CSRC, f)), files))
    return dict(zip(files, res))


def head_commit():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        return None


if __name__ == "__main__":
    table = {"_what": "amdhsa metadata of every kernel (hipcc --offload-arch=gfx950 -O3 -S), per translation unit; private_segment_fixed_size = "
                      "scratch bytes per lane", "_commit": head_commit(), "files": collect()}
    worst = sorted(((k.get("private_segment_fixed_size", 0), f, k["demangled"]) for f, t in table["files"].items() for k in t["kernels"]), reverse=True)[:8]
    table["_largest_scratch"] = [f"{b} B  {f}  {n}" for b, f, n in worst if b]
    txt = json.dumps(table, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")
        print("\n".join(table["_largest_scratch"]) or "no kernel uses scratch")
    else:
        print(txt)
