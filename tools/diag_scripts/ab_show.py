import json,sys
keys = sys.argv[2].split(",") if len(sys.argv) > 2 else ("k_assemble","k_schur","k_solve","k_vis_lin","k_lm_step")
for l in open(sys.argv[1]):
    if " {" in l and (l.startswith("default") or l.startswith("so=")):
        k,j=l.split(" ",1); r=json.loads(j); print(k[-22:], round(r["solves_per_s"]), {a:b for a,b in r["profile_us_per_launch"].items() if a in keys}, r["final_cost"][0])
    elif " vs " in l: print(l.strip()[-60:])
