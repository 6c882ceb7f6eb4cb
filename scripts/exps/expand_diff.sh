#!/bin/bash
# Same hyper-parameters of record as the reference's scripts/exps/expand_diff.sh.  Two calling conventions:
#   sh scripts/exps/expand_diff.sh <EXPAND_NUM>                 one launcher, all GPUs of the node (GPUS=8 by default): the ranks are
#                                                              spawned by generate_data.py --gpus, rank 0 loads the weights once and
#                                                              broadcasts the packed buffers over RCCL (distdiff_amd/launcher.py)
#   sh scripts/exps/expand_diff.sh <EXPAND_NUM> <GPU> <SPLIT>   the reference's positional interface: one process on one GPU,
#                                                              shard SPLIT of TOTAL_SPLIT (single_exp.sh:4-8 style fan-out)
# HIP_VISIBLE_DEVICES replaces CUDA_VISIBLE_DEVICES; MODEL must be a local Hugging Face directory (no network).
SCALE=7.5
DATASET="caltech-101"
START=20
PERIOD=2
CON=0.2
K=3
EXPAND_NUM=$1
GPU=$2
SPLIT=$3
GUIDANCE_TYPE="transform_guidance"
RHO=10.0
STRENGTH=0.5
GUIDE_MODEL="resnet50"
GUIDE_MODEL_WEIGHT="checkpoint/${DATASET}/resnet50_unpretrained_lr0.1/seed1/model_best.pth.tar"
MODEL=${MODEL:-"CompVis/stable-diffusion-v1-4"}
TOTAL_SPLIT=${TOTAL_SPLIT:-4}
GPUS=${GPUS:-8}

DATA_SAVE_PATH=data/${DATASET}_expansion/save/distdiff_batch_${EXPAND_NUM}x
COMMON="--guidance_type=${GUIDANCE_TYPE} -a ${GUIDE_MODEL} -d ${DATASET} --output_dir ${DATA_SAVE_PATH} \
        --gradient_checkpointing --K ${K} --train_batch_size 1 --optimize_targets global_prototype-local_prototype \
        --strength ${STRENGTH} --num_images_per_prompt ${EXPAND_NUM} --guidance_step ${START} --guidance_period ${PERIOD} \
        --encoder_weight_path ${GUIDE_MODEL_WEIGHT} --guidance_scale ${SCALE} --constraint_value ${CON} --rho ${RHO}"
if [ -z "${GPU}" ]; then
    python generate_data.py ${COMMON} --pretrained_model_name_or_path "${MODEL}" --gpus ${GPUS}
else
    HIP_VISIBLE_DEVICES=${GPU} python generate_data.py ${COMMON} --pretrained_model_name_or_path "${MODEL}" \
        --total_split ${TOTAL_SPLIT} --split ${SPLIT}
fi
