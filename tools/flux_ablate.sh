cd $GRAFT_REPO_ROOT
for ab in 0 4 5; do ARTIST_HIP_DEBUG=1 ARTIST_HIP_FLUX_ABLATE=$ab python tools/flux_bench.py 1000 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate $ab', d['crop_pixel_loss_fwd']['ms'], d['crop_pixel_loss_fwd_keep']['ms'])"; done
