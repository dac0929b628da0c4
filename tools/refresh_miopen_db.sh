#!/bin/bash
# Refresh dmm_net_amd/miopen_db with MIOpen's own search results for the TRAINING encoder's shapes (run on an MI355X box):
# the shipped files are copied to a scratch directory, MIOPEN_USER_DB_PATH points there (MIOpen's own variable wins over the
# package's seeding), train_encoder.TrainEncoder(miopen_find=True) runs its warm-up with the search on -- forward and data
# gradient of the 3x3 / 7x7 convolutions at 12 and at 4 frames of 255 x 448 (ResNet-101; BASELINE configs[3]) -- and the
# grown files come back under gpurun_out/miopen_db_new/ to be copied over dmm_net_amd/miopen_db/.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/miopen_db_new; mkdir -p $D
cp $R/dmm_net_amd/miopen_db/* $D/
export MIOPEN_USER_DB_PATH=$D
cd /tmp
for f in 12 4; do
  timeout 1200 python $R/tools/cfg4_probe.py train_find --find --steps 6 --frames $f 2>&1 | grep '^{'
done
wc -l $D/*
