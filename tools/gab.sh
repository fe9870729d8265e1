# developer tool (GPU box): sectors per task chosen by the library against always 16, BASELINE configs 3-5 (reduced) and config 1
for w in "" 16; do
  echo "== DRT_SECTORS_WANT=$w"
  for cfg in "shell 3 28" "galaxy 4 16" "nbody 4 28"; do
    DRT_SECTORS_WANT=$w MAX_IT=2 python tools/run_config.py $cfg 2>&1 | grep -E "precalc_cell|dir_cell|scatt|total"
  done
done
