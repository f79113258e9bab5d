V=godotoceanwaves_amd/csrc/build/variants/pad4.so
for rep in 1 2; do
  echo "== pad 5 (default)"; python scripts/ab_merged.py 1024:4 512:8 2048:1
  echo "== pad 4 (rounds 1-2)"; OCEAN_WAVES_LIB=$V python scripts/ab_merged.py 1024:4 512:8 2048:1
done
