# A/B of the persistent image-facing kernels (k_b2s_tapkp, k_wgrad_tapnp) against the one-shot / sliced ones (GPU box)
for v in 0 1; do echo "oneshot=$v"; PATCHGAN_EXPERIMENT=1 PATCHGAN_TAPK_ONESHOT=$v python tools/layer_bench.py d0/N d0/2N enc0 dec6 2>/dev/null | grep -v "^layer\|sum ms" | cut -c1-150; done
