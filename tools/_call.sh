mkdir -p gpurun_out/r6o
for i in 1 2 3 4; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('run $i: value %.1f ms/step %.4f regions %s pcie %s submit %s call %s call_host %s frac %.4f frac_rocprof %s yolo %.4f conv2 %.5f' % (d['value'], d['ms_per_step'], c['repeat_values']['all'], c['pcie_inclusive_images_per_sec'], c['net_submit_images_per_sec'], c['net_call_images_per_sec'], c['net_call_host_images_per_sec'], d['roofline']['frac'], d['roofline'].get('frac_rocprof'), d['extra']['yolov3_b1']['ms_per_step'], d['extra']['config2']['ms_per_step']))"; done | tee gpurun_out/r6o/driver_form_repeats.txt
