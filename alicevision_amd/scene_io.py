"""Write a synthetic scene (alicevision_amd.synthetic.Scene) as the on-disk input of aliceVision_depthMapEstimation:
an SfMData .sfm (JSON as sfmDataIO/jsonIO.cpp of the reference writes it, version 1.2.11: every scalar a string, rotations
column-major) with pinhole intrinsics, poses and landmarks + observations, and one linear float EXR per view named
<viewId>.exr (what PrepareDenseScene hands to the stage).  SURVEY.md §8d: landmarks are sampled on the analytic surface and
observed in every view that sees them, so the reference's CPU heuristics (>= 21 common landmarks per camera pair, ray angles
in [2, 70] degrees, common landmarks per T camera) are satisfied.
"""
import json
import os

import numpy as np
import torch

from . import exr_io
from .synthetic import _surface

SENSOR_WIDTH = 36.0


def view_id(i):
    return 1000 + 37 * i


def sample_landmarks(scene, n=400, seed=11, z0=4.0, amp=0.2):
    """3-D points on the surface inside the frustum of view 0, with their exact projections in every view that sees them."""
    rng = np.random.RandomState(seed)
    f = scene.K[0, 0]
    half_w, half_h = 0.5 * scene.width / f * z0, 0.5 * scene.height / f * z0
    pts = []
    x = rng.uniform(-0.95 * half_w, 0.95 * half_w, n)
    y = rng.uniform(-0.95 * half_h, 0.95 * half_h, n)
    z = _surface(torch.from_numpy(x), torch.from_numpy(y), z0, amp).numpy()
    X = np.stack([x, y, z], axis=1)
    for k in range(n):
        obs = {}
        for i in range(len(scene.R)):
            pc = scene.R[i] @ (X[k] - scene.C[i])
            if pc[2] <= 0:
                continue
            uv = scene.K @ pc
            u, v = uv[0] / uv[2], uv[1] / uv[2]
            if 0 <= u < scene.width and 0 <= v < scene.height:
                obs[i] = (float(u), float(v))
        if len(obs) >= 2:
            pts.append((X[k], obs))
    return pts


def sfm_dict(scene, landmarks, image_folder=""):
    n = len(scene.R)
    f = float(scene.K[0, 0])
    d = {"version": ["1", "2", "11"], "featuresFolders": [], "matchesFolders": []}
    d["views"] = [{"viewId": str(view_id(i)), "poseId": str(view_id(i)), "frameId": "0", "intrinsicId": "1", "resectionId": "0",
                   "path": os.path.join(image_folder, "%d.exr" % view_id(i)), "width": str(scene.width), "height": str(scene.height),
                   "metadata": {"AliceVision:SensorWidth": "%.6f" % SENSOR_WIDTH}} for i in range(n)]
    d["intrinsics"] = [{"intrinsicId": "1", "width": str(scene.width), "height": str(scene.height), "sensorWidth": "%.17g" % SENSOR_WIDTH,
                        "sensorHeight": "%.17g" % (SENSOR_WIDTH * scene.height / scene.width), "serialNumber": "synthetic", "type": "pinhole",
                        "initializationMode": "calibrated", "initialFocalLength": "-1", "focalLength": "%.17g" % (f * SENSOR_WIDTH / scene.width),
                        "pixelRatio": "1", "pixelRatioLocked": "true", "offsetLocked": "false", "scaleLocked": "false",
                        "principalPoint": ["%.17g" % (scene.K[0, 2] - scene.width / 2.0), "%.17g" % (scene.K[1, 2] - scene.height / 2.0)],
                        "distortionInitializationMode": "none", "distortionParams": [], "undistortionOffset": ["0", "0"], "undistortionParams": [],
                        "distortionType": "none", "undistortionType": "none", "locked": "false"}]
    d["poses"] = [{"poseId": str(view_id(i)),
                   "pose": {"transform": {"rotation": ["%.17g" % v for v in scene.R[i].T.flatten()], "center": ["%.17g" % v for v in scene.C[i]]},
                            "locked": "0"}} for i in range(n)]
    d["structure"] = [{"landmarkId": str(k), "descType": "sift", "color": ["255", "255", "255"], "X": ["%.17g" % v for v in X],
                       "observations": [{"observationId": str(view_id(i)), "featureId": str(k), "x": ["%.17g" % u, "%.17g" % v], "scale": "1"}
                                        for i, (u, v) in sorted(obs.items())]} for k, (X, obs) in enumerate(landmarks)]
    return d


def write_scene(scene, folder, n_landmarks=400, compression=3, with_p_metadata=False):
    """folder/scene.sfm + folder/images/<viewId>.exr; returns (sfm path, images folder)"""
    img_dir = os.path.join(folder, "images")
    os.makedirs(img_dir, exist_ok=True)
    lms = sample_landmarks(scene, n_landmarks)
    sfm = os.path.join(folder, "scene.sfm")
    with open(sfm, "w") as f:
        json.dump(sfm_dict(scene, lms, img_dir), f, indent=1)
    for i in range(len(scene.R)):
        im = scene.images[i].cpu().numpy()
        attrs = {}
        if with_p_metadata:  # what PrepareDenseScene stores (MultiViewParams.cpp:150-156 reads it back)
            P = scene.K @ np.concatenate([scene.R[i], (-scene.R[i] @ scene.C[i])[:, None]], axis=1)
            attrs["AliceVision:P"] = exr_io.m44d(list(P.flatten()) + [0, 0, 0, 1])
            attrs["AliceVision:downscale"] = 1
        exr_io.write_exr(os.path.join(img_dir, "%d.exr" % view_id(i)), {"R": im[..., 0], "G": im[..., 1], "B": im[..., 2], "A": im[..., 3]},
                         attributes=attrs, compression=compression)
    return sfm, img_dir
