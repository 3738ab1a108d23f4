# Shard-sized END-TO-END runs of BASELINE configs[4] and configs[2] on one GPU (VERDICT r4 "next" 3): what one rank of an 8-GPU job
# would do -- eval_MoCoDAD.py's test_step loop + gather + frame-score assembly + AUC -- with the windows generated on device
# (--device-windows) and once more from host-materialised windows.  cfg 4: 1 M windows / 8 = 125 000 (seq_len 24, ns 50, S 8,
# batch 4096); cfg 2: 2.5 M / 8 = 312 500 (seg_len 6, ns 10, S 5, batch 2048).  Synthetic clips: 3 persons x 200 frames x 5
# transforms each, so 48 clips ~ 125 k windows at seg_len 24 and 110 clips ~ 313 k windows at seg_len 6.
#   usage: gpurun -- bash tools/e2e_shards.sh <tag> <git HEAD>
TAG=${1:-r05}
HEAD=${2:-unknown}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
SO=$(sha256sum mocodad_amd/libmocodad_hip.so | cut -c1-64)
BOX=$( (cat /etc/machine-id 2>/dev/null || hostname) | cut -c1-12)
MAN="# manifest: head=$HEAD so_sha256=$SO box=$BOX run_dir=$TAG date=$(date -u +%Y-%m-%dT%H:%MZ)"
{
  echo "$MAN"
  python -c "import torch; torch.zeros(1).cuda()"     # page the image in outside the timed commands
  for rep in 1 2; do
    echo "## cfg 4 shard (configs/seq24_synth.yaml, 48 clips), windows generated on device, run $rep"
    timeout 900 python eval_MoCoDAD.py -c configs/seq24_synth.yaml --synthetic 48 --device-windows --random-init 2>&1 | tail -2
  done
  echo "## cfg 4 shard, host-materialised windows"
  timeout 900 python eval_MoCoDAD.py -c configs/seq24_synth.yaml --synthetic 48 --random-init 2>&1 | tail -2
  for rep in 1 2; do
    echo "## cfg 2 shard (configs/hr_stc_test.yaml, 110 clips), windows generated on device, run $rep"
    timeout 600 python eval_MoCoDAD.py -c configs/hr_stc_test.yaml --synthetic 110 --device-windows --random-init 2>&1 | tail -2
  done
  echo "## cfg 2 shard, host-materialised windows"
  timeout 600 python eval_MoCoDAD.py -c configs/hr_stc_test.yaml --synthetic 110 --random-init 2>&1 | tail -2
  echo "## cfg 1 (configs/hr_avenue_test.yaml, 64 clips), windows generated on device"
  timeout 600 python eval_MoCoDAD.py -c configs/hr_avenue_test.yaml --synthetic 64 --device-windows --random-init 2>&1 | tail -2
} > $O/e2e_shards.txt 2>&1
cat $O/e2e_shards.txt
