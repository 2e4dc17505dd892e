# P = 4 against P = 8 (planar filter build; LSLAM_ICP_VARIANT=1 keeps both on the big build) over problem size, batch size, iterations
for cfg in "2600 1024 20" "3000 1024 20" "3400 1024 20" "3800 1024 20" "4096 512 20" "4096 1024 20" "4096 2048 50" "4096 4096 50" "6000 1024 20" "8192 512 20"; do
  set -- $cfg
  echo "n=$1 S=$2 I=$3"
  C4_P=$1 C4_S=$2 C4_I=$3 python tools/c4_probe.py LSLAM_ICP_POINTS_PER_LANE=4,LSLAM_ICP_VARIANT=1 LSLAM_ICP_POINTS_PER_LANE=8,LSLAM_ICP_VARIANT=1 2>&1 | grep -E "POINTS" | awk '{print "   ", $1, $2, $3}'
done
