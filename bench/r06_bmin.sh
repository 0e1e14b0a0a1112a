cd $GRAFT_REPO_ROOT
for c in C1 C2; do
  python bench/quick.py $c
  for b in 4 12 16 24; do python bench/quick.py $c DBAT_HIP_TILE_BMIN=$b; done
done
