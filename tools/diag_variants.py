import os as _os, sys as _sys
_r = _os.path.dirname(_os.path.abspath(__file__))
while not _os.path.exists(_os.path.join(_r, "_gfbe_import.py")):
    _r = _os.path.dirname(_r)
_sys.path[:0] = [_r, _os.path.join(_r, "tests")]   # (measurement scripts: the package root and the test helpers they share)
"""Kernel variants side by side (one gpurun call instead of one per build): `build` compiles the variants into
ground-fusion2_amd/csrc/variants/ HERE (hipcc cross-compiles without a GPU; the .so files travel with the snapshot),
`run` times every variant on the GPU box in its own process (GFBE_LIB selects the library).
  python tools/diag_variants.py build [names...]      python tools/diag_variants.py run [names...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "ground-fusion2_amd", "csrc", "variants")
VARIANTS = {   # name -> (extra flags, fp-contract)
    "base": ([], "off"),
    "eignoapply": (["-DGFBE_EIG_NOAPPLY=1"], "off"),
    "schurabs": (["-DGFBE_SCHUR_COMPACT=0"], "off"),
    "kvis3": (["-DGFBE_KVIS_WAVES=3"], "off"),
    "densetp0": (["-DGFBE_DENSE_TP=0"], "off"),
    "asmu6": (["-DGFBE_ASM_U=6"], "off"),
    "asmu8": (["-DGFBE_ASM_U=8"], "off"),
    "asmu2": (["-DGFBE_ASM_U=2"], "off"),
    "cand512": (["-DCAND_THREADS=512"], "off"),
    "cand1024": (["-DCAND_THREADS=1024"], "off"),
    "schur3": (["-DGFBE_SCHUR_WGS=3", "-DHS_LD=83"], "off"),
    "noearly": (["-DGFBE_KVIS_EARLY=0"], "off"),
    "contract": ([], "fast"),
    "stamp": (["-DGFBE_KVIS_STAMP=1"], "off"),
    "noesym": (["-DGFBE_SOLVE_ESYM=0"], "off"),
    "cholstamp": (["-DGFBE_CHOL_STAMP=1"], "off"),
    "chainstamp": (["-DGFBE_CHAIN_STAMP=1"], "off"),
    "bigstamp": (["-DGFBE_BIG_STAMP=1"], "off"),
    "ldltstamp": (["-DGFBE_LDLT_STAMP=1"], "off"),
    "clearlm": (["-DGFBE_CLEAR_LM=1"], "off"),
    "linstamp": (["-DGFBE_LIN_STAMP=1"], "off"),
    "linstamp3": (["-DGFBE_LIN_STAMP=1", "-DGFBE_LIN_STAMP_MODE=3"], "off"),
    "linstamp3_512": (["-DGFBE_LIN_STAMP=1", "-DGFBE_LIN_STAMP_MODE=3", "-DGFBE_LIN_SMALL_THREADS=512"], "off"),
    "t256ks4": (["-DGFBE_LIN_SMALL_THREADS=256", "-DGFBE_LIN_SMALL_KS=4"], "off"),
    "linstamp3_t256ks4": (["-DGFBE_LIN_STAMP=1", "-DGFBE_LIN_STAMP_MODE=3", "-DGFBE_LIN_SMALL_THREADS=256", "-DGFBE_LIN_SMALL_KS=4"], "off"),
    "visasm6": (["-DGFBE_VISASM_WAVES=6"], "off"),
    "visasm5": (["-DGFBE_VISASM_WAVES=5"], "off"),
    "visasm3": (["-DGFBE_VISASM_WAVES=3"], "off"),
    "schur5": (["-DGFBE_SCHUR_WGS=5"], "off"),
    "kvis4": (["-DGFBE_KVIS_WAVES=4"], "off"),
    "kvis2": (["-DGFBE_KVIS_WAVES=2"], "off"),
    "chunk8": (["-DVIS_CHUNK=8"], "off"),
    "chunk6": (["-DVIS_CHUNK=6"], "off"),
    "nochunk": (["-DGFBE_VIS_CHUNK=0"], "off"),
    "marg256": (["-DGFBE_MARG_TP_THREADS=256"], "off"),
    "marg1024": (["-DGFBE_MARG_TP_THREADS=1024"], "off"),
    "vb256": (["-DVB_GROUP=256", "-DGFBE_VISASM_WAVES=2"], "off"),
    "vb384": (["-DVB_GROUP=384", "-DGFBE_VISASM_WAVES=3"], "off"),
    "noldlttp": (["-DGFBE_LDLT_TP=0"], "off"),
    "ldlttp2": (["-DGFBE_LDLT_TP=2"], "off"),
    "pf2": (["-DGFBE_WIDE_PREFETCH2=1"], "off"),
    "nosimdroles": (["-DGFBE_CHAIN_SIMD_ROLES=0"], "off"),
    "ql": (["-DGFBE_EIG_DC=0"], "off"),
    "lmsstamp": (["-DGFBE_LMS_STAMP=1"], "off"),
    "linstamp1": (["-DGFBE_LIN_STAMP=1", "-DGFBE_LIN_STAMP_MODE=1"], "off"),
    "lin512": (["-DGFBE_LIN_SMALL_THREADS=512"], "off"),
    "chain1wg": (["-DGFBE_CHAIN_LDS_PAD=24576"], "off"),
    "asmtp0": (["-DGFBE_ASM_TP=0"], "off"),
    "asmtp1": (["-DGFBE_ASM_TP=1"], "off"),
    "asmtp5": (["-DGFBE_ASM_TP=5"], "off"),
    "asmtp13": (["-DGFBE_ASM_TP=13"], "off"),
    "margaside0": (["-DGFBE_MARG_DENSE_ASIDE=0"], "off"),
    "pcs0": (["-DGFBE_PCS_ONE_ROUND=0"], "off"),
    "stepcand0": (["-DGFBE_STEP_CAND_REGS=0"], "off"),
    "lmsstamp0": (["-DGFBE_LMS_STAMP=1", "-DGFBE_LMS_AHEAD=0"], "off"),
    "lmsa0": (["-DGFBE_LMS_AHEAD=0"], "off"),
    "lin1024": (["-DGFBE_LIN_SMALL_THREADS=1024"], "off"),
    "lin512ks5": (["-DGFBE_LIN_SMALL_THREADS=512", "-DGFBE_LIN_SMALL_KS=5"], "off"),
    "fuse0": (["-DGFBE_FUSE_SMALL=0"], "off"),
    "fuse1": (["-DGFBE_FUSE_SMALL=1"], "off"),
    "fuse3": (["-DGFBE_FUSE_SMALL=3"], "off"),
    "fuse5": (["-DGFBE_FUSE_SMALL=5"], "off"),
    "ks5": (["-DGFBE_LIN_SMALL_KS=5"], "off"),
    "ks10": (["-DGFBE_LIN_SMALL_KS=10"], "off"),
    "ks4": (["-DGFBE_LIN_SMALL_KS=4"], "off"),
    "ks6": (["-DGFBE_LIN_SMALL_KS=6"], "off"),
    "ks7": (["-DGFBE_LIN_SMALL_KS=7"], "off"),
    "s512": (["-DSOLVE_THREADS=512", "-DSOLVE_WAVES_PER_EU=2"], "off"),
    "s512u10": (["-DSOLVE_THREADS=512", "-DSOLVE_WAVES_PER_EU=2", "-DBUILD_UNROLL=10"], "off"),
    "s512u13": (["-DSOLVE_THREADS=512", "-DSOLVE_WAVES_PER_EU=2", "-DBUILD_UNROLL=13"], "off"),
    "s768": (["-DSOLVE_THREADS=768", "-DSOLVE_WAVES_PER_EU=3", "-DBUILD_UNROLL=9"], "off"),
    "s768i": (["-DSOLVE_THREADS=768", "-DSOLVE_WAVES_PER_EU=3", "-DBUILD_UNROLL=9", "-DGFBE_SOLVE_INLINE=1"], "off"),
    "s768u6": (["-DSOLVE_THREADS=768", "-DSOLVE_WAVES_PER_EU=3", "-DBUILD_UNROLL=6"], "off"),
    "s512i": (["-DSOLVE_THREADS=512", "-DSOLVE_WAVES_PER_EU=2", "-DGFBE_SOLVE_INLINE=1"], "off"),
    "cprio1": (["-DGFBE_CHAIN_PRIO=1"], "off"),
    "cprio3": (["-DGFBE_CHAIN_PRIO=3"], "off"),
    "sprio": (["-DGFBE_PRIO_SMALL=1"], "off"),
    "allprio": (["-DGFBE_PRIO_SMALL=1", "-DGFBE_CHAIN_PRIO=3"], "off"),
    "nostepcand": (["-DGFBE_FUSE_STEP_CAND=0"], "off"),
    "abl1_nomfma": (["-DGFBE_ABLATE=1"], "off"),
    "abl2_nopartstore": (["-DGFBE_ABLATE=2"], "off"),
    "abl3_nohpstore": (["-DGFBE_ABLATE=3"], "off"),
    "abl4_noeval": (["-DGFBE_ABLATE=4"], "off"),
}

def build(names):
    from _gfbe_import import gf
    os.makedirs(VDIR, exist_ok=True)
    for n in names:
        flags, fc = VARIANTS[n]
        print("build", n, flags, fc, flush=True)
        gf.build_native(force=True, out=os.path.join(VDIR, "libgfbe_%s.so" % n), extra_flags=flags + ["-DGFBE_DIAG=1"], fp_contract=fc)

def run(names):
    for n in names:
        so = os.path.join(VDIR, "libgfbe_%s.so" % n)
        if not os.path.exists(so):
            print("== %s: not built" % n); continue
        env = dict(os.environ, GFBE_LIB=so, B=os.environ.get("B", "1024"), SPLIT=os.environ.get("SPLIT", "1"))
        print("== variant %s" % n, flush=True)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diag_kvis.py")], env=env, capture_output=True, text=True, timeout=600)
        print("\n".join(l for l in r.stdout.splitlines() if l.split() and l.split()[0] in ("k_vis_lin_iter0", "k_vis_lin", "k_schur", "k_solve", "k_assemble", "k_visblock", "solves_per_s", "final_cost")))
        if r.returncode != 0:
            print("rc", r.returncode, r.stderr[-600:])

if __name__ == "__main__":
    names = sys.argv[2:] or list(VARIANTS)
    {"build": build, "run": run}[sys.argv[1]](names)
