#!/usr/bin/env python3
"""Stage-1 micro-server — mirrors multi_target_lip2speech/inference_server.py:229-384 on the HIP path.

Same three routes and status codes:
  GET  /checkpoints                       -> JSON list of checkpoint ids                       (:229-231)
  POST /load_checkpoint {checkpoint_id}   -> 204 (also when already loaded) | 400 unknown id   (:233-248)
  POST /synthesise                        -> 204; re-reads <label_dir>/{test.tsv,test.unt} and writes pred_unit/, pred_mel/,
                                             hypo-<fid>.json, wer.<fid> under common_eval.results_path (:250-384)
started like the reference (`start_server.sh:30`):
  python -m lip2speech_unit_amd.inference_server override.checkpoints_data_path=<checkpoints.json> \
      common_eval.results_path=<dir> override.data=<label_dir> override.label_dir=<label_dir> [port=5004] [fp16=true]
checkpoints.json = {"default_checkpoint_id": "<id>", "checkpoints": {"<id>": "<path.pt>", ...}} (:203-209).

Differences, all on purpose: checkpoints are read from disk when selected instead of all being held on the CPU (:106-114);
clips are batched (dataset.batch_size) with clip-alone results; with `vocoder.config` set, /synthesise hands units + mel to
the vocoder in device memory and writes pred_wav/ too, so the service needs no second server for stage 2; and the web
layer's rule that sends clips longer than MAX_GPU_DURATION = 10 s to a CPU copy of this server (server.py:288,
start_server.sh:33-39) is unnecessary here: one MI355X takes the service's 24-s limit (config.py:30) in a single batch.
Like the reference the app keeps module-level state (:47); the two routes that touch the GPU serialise on one lock.
"""
import gc
import json
import sys
import threading
from http import HTTPStatus

import torch

from . import inference as s1
from .task import Lip2SpeechTask, decode_config

# One request at a time touches the GPU state: the generator's captured hipGraphs replay into static buffers, and a model
# switch frees what a running replay reads.  Flask serves threaded by default, so both routes take this lock.
gpu_lock = threading.Lock()
state = {"model": None, "loaded_checkpoint_id": None, "task": None, "vocoder": None, "sampling_rate": 16000, "generator": None}


def switch_model(checkpoint_id, checkpoints, cfg, logger):
    """inference_server.py:152-175: release the previous model, bring the selected one to the GPU."""
    logger.info(f"SWITCHING MODEL: {checkpoint_id}")
    if state["model"] is not None:
        state["model"] = state["generator"] = None      # the generator holds the model and its captured hipGraphs
        gc.collect()
        torch.cuda.empty_cache()
    state["model"] = s1.build_model(cfg, state["task"], logger, checkpoint_path=checkpoints[checkpoint_id])
    state["loaded_checkpoint_id"] = checkpoint_id


def create_app(cfg):
    from flask import Flask, request
    if not torch.cuda.is_available():
        raise SystemExit("this build runs on MI355X only: no CPU path")
    results_path = cfg["common_eval.results_path"]
    assert results_path, "common_eval.results_path is required"
    logger = s1.setup_logging(results_path)
    with open(cfg["override.checkpoints_data_path"]) as f:
        checkpoints_data = json.load(f)
    checkpoints = checkpoints_data["checkpoints"]
    tcfg = decode_config(data=cfg["override.data"], label_dir=cfg["override.label_dir"], fp16=bool(cfg["fp16"]))
    state["task"] = Lip2SpeechTask(tcfg)
    if cfg["vocoder.config"]:
        state["vocoder"], h = s1.build_vocoder(cfg)
        state["sampling_rate"] = h.get("sampling_rate", 16000)
    switch_model(checkpoints_data["default_checkpoint_id"], checkpoints, cfg, logger)

    app = Flask(__name__)

    @app.get("/checkpoints")
    def get_checkpoints():
        return list(checkpoints.keys())

    @app.post("/load_checkpoint")
    def load_checkpoint():
        checkpoint_id = request.json["checkpoint_id"]
        if checkpoint_id == state["loaded_checkpoint_id"]:
            return "", HTTPStatus.NO_CONTENT
        if not checkpoints.get(checkpoint_id):
            return {"message": f"Checkpoint '{checkpoint_id}' does not exist"}, HTTPStatus.BAD_REQUEST
        with gpu_lock:
            switch_model(checkpoint_id, checkpoints, cfg, logger)
        return "", HTTPStatus.NO_CONTENT

    @app.post("/synthesise")
    def synthesise():
        task = state["task"]
        with gpu_lock:
            ds = task.load_dataset(cfg["dataset.gen_subset"])     # manifests are re-read per request (:252)
            if state["generator"] is None:                     # one per loaded checkpoint: its hipGraphs outlive the request
                state["generator"] = s1.build_generator(cfg, task, state["model"], results_path)
            s1.decode_dataset(cfg, task, state["model"], ds, results_path, logger, vocoder=state["vocoder"],
                              sampling_rate=state["sampling_rate"], generator=state["generator"])
        return "", HTTPStatus.NO_CONTENT

    return app


def main(argv=None):
    cfg = s1.parse_overrides(sys.argv[1:] if argv is None else argv)
    cfg.setdefault("override.checkpoints_data_path", None)
    assert cfg["override.checkpoints_data_path"], "override.checkpoints_data_path is required"
    create_app(cfg).run(port=int(cfg.get("port", 5004)))


if __name__ == "__main__":
    main()
