# A/B of the persistent image-facing kernel k_b2s_tapkp against the one-shot k_b2s_tapk, and its workgroup count (GPU box)
for v in "0 512" "0 256" "0 768" "0 1024" "1 512"; do set -- $v; echo "oneshot=$1 wg=$2"; PATCHGAN_EXPERIMENT=1 PATCHGAN_TAPK_ONESHOT=$1 PATCHGAN_TAPKP_WG=$2 python tools/layer_bench.py d0/N d0/2N enc0 dec6 2>/dev/null | grep -v "^layer\|sum ms" | cut -c1-80; done
