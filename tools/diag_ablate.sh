for a in ${ABL:-0 1 2}; do
  GFBE_EXTRA_FLAGS="-DGFBE_ABLATE=$a" python -c "
from _gfbe_import import gf
gf.build_native(force=True)" 2>&1 | grep -E "error" ; 
  echo "== ablate $a"; python tools/diag_kvis.py 2>&1 | grep -E "k_vis_lin_iter0|k_vis_lin "
done
