for t in 4 2; do echo "== OW_DEBUG_PAIR_TEXELS=$t"; OW_DEBUG_PAIR_TEXELS=$t python scripts/ab_merged.py 1024:8 1024:4 1024:6 512:8; done
