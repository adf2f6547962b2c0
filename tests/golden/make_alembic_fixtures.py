"""Fixtures of the Alembic (.abc) SfMData reader: the reference's own compatibility scenes.

Run where /root/reference exists:  python tests/golden/make_alembic_fixtures.py

sfmDataIO/compatibilityData of the reference holds one sample scene (sceneSample.cpp: 54 views, 2 intrinsics, 27 poses, 9261
landmarks) saved by nine versions of its IO code, each as .abc and as .json (sfmDataIOCompatibility_test.cpp loads every one and
compares it with the generated scene).  This script
  * copies five of the .abc files, gzip-ed, into tests/golden/alembic/ — the versions either side of every rule the importer
    switches on (AlembicImporter.cpp: < 1.2.1 principal point relative to the corner, < 1.2.3 no graphics <-> vision flip,
    < 1.2.8 distortion named by the intrinsic type) and the newest;
  * writes expected.json from the NEWEST .json twin with nothing but the json module: view ids, the two intrinsics in pixels
    (jsonIO.cpp:302-346 for version 1.2.11: fx = focalLength / pixelRatio * width / sensorWidth, fy = focalLength * width /
    sensorWidth, principal point = offset from the image centre), poses (rotation stored column-major, jsonIO.hpp:49-63),
    the landmark count and every 97th landmark.
tests/test_host_cpu.py::test_alembic_* read the archives with the C++ reader and compare.
"""
import gzip
import json
import os
import shutil

SRC = "/root/reference/src/aliceVision/sfmDataIO/compatibilityData"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "alembic")
VERSIONS = ["1.2.0", "1.2.2", "1.2.3", "1.2.8", "1.2.11"]


def main():
    os.makedirs(DST, exist_ok=True)
    for v in VERSIONS:
        with open(os.path.join(SRC, "scene_v%s.abc" % v), "rb") as f, gzip.GzipFile(os.path.join(DST, "scene_v%s.abc.gz" % v), "wb", 9, mtime=0) as g:
            shutil.copyfileobj(f, g)
    d = json.load(open(os.path.join(SRC, "scene_v1.2.11.json")))
    assert d["version"] == ["1", "2", "11"]
    exp = {"views": [], "intrinsics": [], "poses": [], "n_landmarks": len(d["structure"]), "landmarks": []}
    for v in d["views"]:
        exp["views"].append({"viewId": int(v["viewId"]), "poseId": int(v["poseId"]), "intrinsicId": int(v["intrinsicId"]), "path": v["path"],
                             "width": int(v["width"]), "height": int(v["height"]), "metadata": dict(v.get("metadata", {}))})
    for i in d["intrinsics"]:
        w, sw = float(i["width"]), float(i["sensorWidth"])
        f, par = float(i["focalLength"]), float(i["pixelRatio"])
        exp["intrinsics"].append({"intrinsicId": int(i["intrinsicId"]), "type": i["type"], "distortionType": i["distortionType"],
                                  "width": int(i["width"]), "height": int(i["height"]), "sensorWidth": sw, "sensorHeight": float(i["sensorHeight"]),
                                  "scale": [f / par * w / sw, f * w / sw], "offset": [float(x) for x in i["principalPoint"]],
                                  "distortionParams": [float(x) for x in (i["distortionParams"] or [])]})
    for p in d["poses"]:
        r = [float(x) for x in p["pose"]["transform"]["rotation"]]
        exp["poses"].append({"poseId": int(p["poseId"]), "rotation": [r[3 * c + rr] for rr in range(3) for c in range(3)],  # row-major
                             "center": [float(x) for x in p["pose"]["transform"]["center"]]})
    for k, l in enumerate(d["structure"]):
        assert int(l["landmarkId"]) == k and not l["observations"]
        if k % 97 == 0:
            exp["landmarks"].append({"id": k, "X": [float(x) for x in l["X"]]})
    with open(os.path.join(DST, "expected.json"), "w") as f:
        json.dump(exp, f, indent=0, sort_keys=True)
    print("wrote", sorted(os.listdir(DST)))


if __name__ == "__main__":
    main()
