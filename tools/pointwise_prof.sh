#!/bin/bash
# Run ON THE GPU BOX: achieved GB/s of the stand-alone HBM-bound layer kernels from one rocprofv3 kernel trace
#   tools/pointwise_prof.sh r04   ->  gpurun_out/prof/r04_pointwise_gbs.md  (copy into profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r04}
out=$R/gpurun_out/prof
mkdir -p $out/pw
cd /tmp && export TMPDIR=/tmp
python $R/tools/pointwise_gbs.py > $out/pw/hip_events.md 2> $out/pw/hip_events.err
rocprofv3 --kernel-trace -d $out/pw -o pw --output-format csv -- python $R/tools/pointwise_gbs.py --manifest $out/pw/manifest.json > $out/pw/traced.md 2> $out/pw/traced.err
f=$out/${tag}_pointwise_gbs.md
{
  echo "# Stand-alone HBM-bound layer kernels: achieved GB/s (MI355X, fp32, NCHW operator API)"
  echo
  echo "\`tools/pointwise_prof.sh $tag\`: the cases of \`tools/pointwise_gbs.py\` (config 1 CustomNet shapes, ResNet-18 at batch 32, YOLO-v3 at"
  echo "batch 1), 103 launches each.  Algorithmic bytes = inputs read once + output written once; peak 8 TB/s (achievable copy rate 6.3)."
  echo "These are the kernels \`Net.forward\` runs (one per layer, the reference's execution model); compiled plans fold BatchNorm /"
  echo "ReLU / LeakyReLU / Add into the convolution epilogues.  Tensors up to 100 MB fit the 256 MB Infinity Cache: a kernel whose"
  echo "operands were just written finds part of them there."
  echo
  echo "## rocprofv3 --kernel-trace (device time per call = all dispatches of the case / launches)"
  echo
  python $R/tools/pointwise_digest.py $(ls $out/pw/*kernel_trace.csv | head -1) $out/pw/manifest.json
  echo
  echo "## HIP events, untraced (best of 5 bursts of 20 back-to-back calls)"
  echo
  cat $out/pw/hip_events.md
} > $f
tail -40 $f
