"""IoU / NoC metrics (reference core/inference/utils.py:107-146)."""
from typing import List, Tuple

import numpy as np


def get_iou(gt_mask: np.ndarray, pred_mask: np.ndarray, ignore_label: int = -1) -> float:
    keep = gt_mask != ignore_label
    obj = gt_mask == 1
    intersection = np.logical_and(np.logical_and(pred_mask, obj), keep).sum()
    union = np.logical_and(np.logical_or(pred_mask, obj), keep).sum()
    return intersection / union


def compute_noc_metric(all_ious: List[np.ndarray], iou_thrs: List[float], max_clicks: int = 20
                       ) -> Tuple[List[float], List[float], List[int]]:
    def noc(iou_arr, thr):
        hit = iou_arr >= thr
        return np.argmax(hit) + 1 if np.any(hit) else max_clicks

    noc_list, noc_std, over_max = [], [], []
    for thr in iou_thrs:
        scores = np.array([noc(a, thr) for a in all_ious], dtype=np.int_)
        noc_list.append(scores.mean())
        noc_std.append(scores.std())
        over_max.append((scores == max_clicks).sum())
    return noc_list, noc_std, over_max


def get_zoom_in_params(eval_mode: str, dataset_name: str = "") -> dict:
    """Zoom-in parameters of an evaluation mode (core/inference/utils.py:288-316, the ``eval_ritm=False`` branch that
    every experiment of the reference used): ``fixed<H>`` / ``fixed<H>,<W>`` -> zoom from the first click to H x W (W = H
    when omitted); ``cvpr`` -> 448 x 448, DAVIS 672 x 672.  Anything else raises NotImplementedError, as the reference does."""
    eval_mode = str(eval_mode)
    if eval_mode == "cvpr":
        return {"skip_clicks": -1, "target_size": (672, 672) if dataset_name == "DAVIS" else (448, 448)}
    if eval_mode.startswith("fixed"):
        parts = eval_mode.split(",")
        h = int(parts[0][5:])
        return {"skip_clicks": -1, "target_size": (h, int(parts[1]) if len(parts) == 2 else h)}
    raise NotImplementedError(f"eval_mode={eval_mode!r}: 'cvpr', 'fixed<number>' or 'fixed<number>,<number>'")


def get_time_metrics(all_ious: List[np.ndarray], elapsed_time: float) -> Tuple[float, float]:
    """Seconds per click / per image (core/inference/utils.py:149-161)."""
    return elapsed_time / sum(map(len, all_ious)), elapsed_time / len(all_ious)


def get_results_table(noc_list, over_max_list, brs_type, dataset_name, mean_spc, elapsed_time, iou_first, n_clicks=20,
                      model_name=None, upsampler_type=None, single_model_eval=True):
    """The evaluation table of the reference, column for column (core/inference/utils.py:174-246): returns
    (header, row, metrics dict).  Log files written from it are interchangeable with the reference's."""
    from datetime import timedelta
    up_row = f"{upsampler_type:^20}|" if upsampler_type is not None else f'{"":^20}|'
    first_col = f'{"BRS Type":^13}|' if single_model_eval else f'{"Ckpt":^13}|'
    table_header = (f'|{"Upsampler Type":^20}|' + first_col + f'{"Dataset":^11}|{"NoC@80%":^9}|{"NoC@85%":^9}|{"NoC@90%":^9}|'
                    f'{"IoU@1":^9}|{">=" + str(n_clicks) + "@85%":^9}|{">=" + str(n_clicks) + "@90%":^9}|{"SPC,s":^7}|{"Time":^9}|')
    width = len(table_header)
    header = f"Eval results for model: {model_name}\n" if single_model_eval and model_name is not None else ""
    header += "-" * width + "\n" + table_header + "\n" + "-" * width
    eval_time = str(timedelta(seconds=int(elapsed_time)))
    q = f'{"?":^9}|'
    row = f"|{up_row}{brs_type:^13}|{dataset_name:^11}|{noc_list[0]:^9.2f}|"
    row += f"{noc_list[1]:^9.2f}|" if len(noc_list) > 1 else q
    row += f"{noc_list[2]:^9.2f}|" if len(noc_list) > 2 else q
    row += f"{iou_first:^9.2f}|"
    row += f"{over_max_list[1]:^9}|" if len(noc_list) > 1 else q
    row += f"{over_max_list[2]:^9}|" if len(noc_list) > 2 else q
    row += f"{mean_spc:^7.3f}|{eval_time:^9}|"
    metrics = {"NoC@80%": noc_list[0], "NoC@85%": noc_list[1] if len(noc_list) > 1 else -1,
               "NoC@90%": noc_list[2] if len(noc_list) > 2 else -1,
               f">={n_clicks}@85%": over_max_list[1] if len(noc_list) > 1 else -1,
               f">={n_clicks}@90%": over_max_list[2] if len(noc_list) > 2 else -1, "SPC,s": mean_spc, "Time": eval_time}
    return header, row, metrics


def save_results(upsampler_type, dataset_name, logs_path, dataset_results, eval_mode="fixed224", mode="NoBRS", n_clicks=20,
                 target_iou=0.90, print_ious=True, logs_prefix="", row_name="NoBRS", save_ious=False, print_header=True,
                 single_model_eval=True):
    """core/inference/utils.py:365-505 without the hydra config object: prints the table row (+ the per-click mIoU list
    when print_ious), appends it to ``<logs_path>/<prefix><eval_mode>_<mode>_<n_clicks>.txt`` and optionally pickles the
    per-object IoU arrays to ``<logs_path>/ious/<prefix>/<dataset>_<eval_mode>_<mode>_<n_clicks>.pkl``."""
    import pickle
    from pathlib import Path
    logs_path = Path(logs_path)
    all_ious, elapsed = dataset_results
    mean_spc, _ = get_time_metrics(all_ious, elapsed)
    thrs = np.arange(0.8, min(0.95, target_iou) + 0.001, 0.05).tolist()
    noc, _, over = compute_noc_metric(all_ious, iou_thrs=thrs, max_clicks=n_clicks)
    iou_first = np.array([x[0] for x in all_ious]).mean(0)
    model_name = logs_path.stem if not logs_prefix else f"{logs_path.name}:{logs_prefix}"
    header, row, results = get_results_table(noc, over, row_name, dataset_name, mean_spc, elapsed, iou_first, n_clicks,
                                             model_name, upsampler_type, single_model_eval)
    if print_ious:
        k = min(len(x) for x in all_ious)
        mean_ious = np.array([x[:k] for x in all_ious]).mean(axis=0)
        row += "; " + " ".join(f"mIoU@{c}={mean_ious[c - 1]:.2%};" for c in range(1, 21) if c <= k)
        pct = [round(float(v) * 100, 2) for v in mean_ious]
        results.update({f"mIoU@{c}": pct[c - 1] for c in range(1, 21) if c <= k})
        results["miou_list"] = [pct[c - 1] for c in range(1, 21) if c <= k]
        results["clicks_list"] = [c for c in range(1, 21) if c <= k]
    if print_header:
        print(header)
    print(row)
    if save_ious:
        d = logs_path / "ious" / (logs_prefix or "")
        d.mkdir(parents=True, exist_ok=True)
        with open(d / f"{dataset_name}_{eval_mode}_{mode}_{n_clicks}.pkl", "wb") as fp:
            pickle.dump(all_ious, fp)
    prefix = (logs_prefix + "_" + ("" if single_model_eval else f"{dataset_name}_")) if logs_prefix else ""
    logs_path.mkdir(parents=True, exist_ok=True)
    log = logs_path / f"{prefix}{eval_mode}_{mode}_{n_clicks}.txt"
    if log.exists():
        with open(log, "a") as f:
            f.write(row + "\n")
    else:
        with open(log, "w") as f:
            if print_header:
                f.write(header + "\n")
            f.write(row + "\n")
    return results


def save_iou_analysis_data(dataset_name, logs_path, dataset_results, eval_mode="fixed224", mode="NoBRS", n_clicks=20,
                           logs_prefix="", model_name=None):
    """The pickle plot_iou_vs_clicks consumes (core/inference/utils.py:508-543):
    ``<logs_path>/plots/<prefix><dataset>_<eval_mode>_<mode>_<n_clicks>.pickle`` = {dataset_name, model_name, all_ious}."""
    import pickle
    from pathlib import Path
    logs_path = Path(logs_path)
    all_ious, _ = dataset_results
    prefix = (logs_prefix + "_" if logs_prefix else "") + dataset_name + "_"
    if model_name is None:
        model_name = logs_path.stem if not logs_prefix else f"{logs_path.name}:{logs_prefix}"
    path = logs_path / "plots" / f"{prefix}{eval_mode}_{mode}_{n_clicks}.pickle"
    path.parent.mkdir(parents=True, exist_ok=True)
    with path.open("wb") as f:
        pickle.dump({"dataset_name": dataset_name, "model_name": f"{model_name}_{mode}", "all_ious": all_ious}, f)
    return path


def get_save_feats_callback(logs_path, dataset_name, save_folder_name="features", exec_for_n_imgs=10):
    """``feats_callback`` of ``evaluate_sample`` that dumps the raw features of the first click of the first ``exec_for_n_imgs``
    images (reference inference/utils.py:587-627, eval_cfg.yaml ``save_feats``): ``<logs>/feats/<dataset>/<folder>_<time>/
    <sample>_<click>_{LowRes,HighRes}.pth`` -- here contiguous fp32 NCHW CPU tensors, whatever layout / dtype the HIP path held
    them in -- and ``images/<sample>_<click>_image.jpg``, the image with the clicks drawn (PIL; the reference draws with OpenCV)."""
    from datetime import datetime
    from pathlib import Path

    import torch
    save_path = Path(logs_path) / "feats" / dataset_name / f"{save_folder_name}_{datetime.now().strftime('%Y-%m-%d_%H:%M')}"
    (save_path / "images").mkdir(parents=True, exist_ok=True)

    def callback(image, feats, sample_id, click_indx, clicks_list):
        if sample_id >= exec_for_n_imgs or click_indx >= 1:
            return None
        from PIL import Image, ImageDraw

        from ..model._tensor import to_nchw_f32
        for k, v in feats.items():
            torch.save(to_nchw_f32(v).cpu(), str(save_path / f"{sample_id}_{click_indx}_{k}.pth"))
        if isinstance(image, dict):
            image = image["image"]
        pic = Image.fromarray(np.ascontiguousarray(image).astype(np.uint8))
        draw = ImageDraw.Draw(pic)
        for click in clicks_list or ():
            y, x = click.coords
            colour = (0, 255, 0) if click.is_positive else (255, 0, 0)
            draw.ellipse((x - 6, y - 6, x + 6, y + 6), fill=colour, outline=colour)
        pic.save(str(save_path / "images" / f"{sample_id}_{click_indx}_image.jpg"))

    callback.save_path = save_path
    return callback


def load_single_is_model(state_dict, device, eval_ritm=False, **kwargs):
    """core/inference/utils.py:60-83: rebuild the model from the checkpoint's config, load the saved (trainable) weights
    over the freshly constructed ones, freeze, move, eval.  ``eval_ritm`` keeps the reference's positional slot
    (RITM-style evaluation is outside the probed path: True raises)."""
    from ..utils.serialization import load_model
    model = load_model(state_dict["config"], eval_ritm, **kwargs)
    current = model.state_dict()
    current.update(state_dict["state_dict"])
    model.load_state_dict(current, strict=False)
    for p in model.parameters():
        p.requires_grad = False
    return model.to(device).eval()


def load_is_model(checkpoint, device, eval_ritm=False, **kwargs):
    """core/inference/utils.py:37-57: a checkpoint path / dict, or a list of them (per-click models); same positional
    signature as the reference (``load_is_model(ckpt, device, cfg.eval_ritm)``, evaluate.py:80)."""
    import torch
    from pathlib import Path
    import isegprobe_amd
    isegprobe_amd.install_as_core()  # checkpoints name core.* classes and pickle core.utils.model_builder.ModelBuilder
    sd = torch.load(checkpoint, map_location="cpu", weights_only=False) if isinstance(checkpoint, (str, Path)) else checkpoint
    if isinstance(sd, list):
        return (load_single_is_model(sd[0], device, eval_ritm, **kwargs),
                [load_single_is_model(x, device, eval_ritm, **kwargs) for x in sd])
    return load_single_is_model(sd, device, eval_ritm, **kwargs)
