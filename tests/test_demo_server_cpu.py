"""Demo back end: the request/response contract of the reference's demo/app.py routes, exercised over real
HTTP against a stand-in predictor (host logic only; the GPU test drives the HIP predictor)."""
import http.client
import json
import os
import threading

import numpy as np
import pytest
import torch

from point_sam_amd.demo_server import DemoSession, serve


class FakePredictor:
    """mask logits = -(distance to the mean positive prompt) + 0.5 (+ previous logits * 0.1); three identical candidates
    with scores favouring candidate 1 on the first click."""

    def __init__(self):
        self.calls = []
        self.fail_next = False

    def set_pointcloud(self, xyz, rgb):
        self.xyz = xyz

    def predict_masks(self, pts, labels, prompt_mask, multimask):
        if self.fail_next:
            self.fail_next = False
            raise ValueError("predictor refused the prompt")
        self.calls.append((pts.clone(), labels.clone(), None if prompt_mask is None else prompt_mask.clone(), multimask))
        c = pts[0][labels[0].bool()].mean(0) if labels[0].bool().any() else pts[0].mean(0)
        logit = 0.5 - (self.xyz[0] - c).norm(dim=-1)
        if prompt_mask is not None:
            logit = logit + 0.1 * prompt_mask[0]
        C = 3 if multimask else 1
        logits = torch.stack([logit + 0.01 * i for i in range(C)])[None]
        scores = torch.tensor([[0.1, 0.9, 0.3][:C]])
        return logits, scores, logits


@pytest.fixture()
def server(tmp_path):
    pts = np.random.RandomState(0).rand(50, 3) * 4 + 1
    ply = tmp_path / "toy.ply"
    with open(ply, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\n"
                "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
        for p in pts:
            f.write(f"{p[0]} {p[1]} {p[2]} 255 0 128\n")
    pred = FakePredictor()
    sess = DemoSession(pred, models_dir=str(tmp_path), output_dir=str(tmp_path / "results"), device="cpu")
    srv = serve(sess, "127.0.0.1", 0)
    th = threading.Thread(target=srv.serve_forever, daemon=True)
    th.start()
    yield srv.server_address[1], sess, pred, pts, tmp_path
    srv.shutdown()


def _req(port, method, path, body=None):
    c = http.client.HTTPConnection("127.0.0.1", port, timeout=10)
    c.request(method, path, None if body is None else json.dumps(body), {"Content-Type": "application/json"})
    r = c.getresponse()
    return r.status, json.loads(r.read()), dict(r.getheaders())


def test_routes_follow_the_reference_contract(server):
    port, sess, pred, pts, tmp = server
    st, out, hdr = _req(port, "GET", "/pointcloud/toy.ply")
    assert st == 200 and hdr["Access-Control-Allow-Origin"] == "*"
    xyz = np.array(out["xyz"]).reshape(-1, 3)
    want = (pts - pts.mean(0)) / np.linalg.norm(pts - pts.mean(0), axis=-1).max()
    assert np.allclose(xyz, want) and np.allclose(np.array(out["rgb"]).reshape(-1, 3), [1.0, 0.0, 128 / 255])
    assert abs(np.linalg.norm(xyz, axis=-1).max() - 1) < 1e-12                       # unit ball (app.py:123-126)
    # first click: multimask, best-score candidate (index 1) returned and kept as the next dense prompt
    st, out, _ = _req(port, "POST", "/segment", {"prompt_point": xyz[3].tolist(), "prompt_label": 1})
    assert st == 200 and len(out["seg"]) == 50 and out["seg"][3] is True
    assert pred.calls[0][2] is None and pred.calls[0][3] is True
    # second click: both prompts, previous logits as prompt_mask, single-mask output
    st, out2, _ = _req(port, "POST", "/segment", {"prompt_point": xyz[7].tolist(), "prompt_label": 0})
    pts2, lab2, pm2, multi2 = pred.calls[1]
    assert pts2.shape == (1, 2, 3) and lab2.tolist() == [[1, 0]] and multi2 is False and pm2.shape == (1, 50)
    assert torch.equal(pm2[0], 0.5 - (sess.pc_xyz[0] - sess.pc_xyz[0][3]).norm(dim=-1) + 0.01)   # candidate 1 of click 1
    # next keeps the mask and clears the prompts; save writes the reference's npy dict
    assert _req(port, "POST", "/next")[1] == {"status": "cleared"} and sess.prompts == [] and len(sess.masks) == 1
    _req(port, "POST", "/segment", {"prompt_point": xyz[1].tolist(), "prompt_label": 1})
    assert pred.calls[2][3] is True                                                  # fresh object: multimask again
    assert _req(port, "POST", "/clear")[1] == {"status": "cleared"} and sess.prompt_mask is None
    assert _req(port, "POST", "/save")[1] == {"status": "saved"}
    saved = np.load(tmp / "results" / "toy.npy", allow_pickle=True).item()
    assert saved["mask"].shape == (1, 50) and saved["xyz"].shape == (50, 3) and np.array_equal(saved["mask"][0], np.array(out2["seg"]))


def test_sampled_pointcloud_and_errors(server):
    port, sess, pred, pts, _ = server
    st, out, _ = _req(port, "POST", "/segment", {"prompt_point": [0, 0, 0], "prompt_label": 1})
    assert st == 400 and "before a point cloud" in out["error"]
    flat = {str(i): float(v) for i, v in enumerate(np.linspace(-0.5, 0.5, 30))}
    st, out, _ = _req(port, "POST", "/sampled_pointcloud", {"points": flat, "colors": flat})
    assert st == 200 and out == {"response": "success"} and sess.pc_xyz.shape == (1, 10, 3)
    assert _req(port, "POST", "/next")[0] == 400                       # nothing segmented yet
    assert _req(port, "POST", "/nope")[0] == 404 and _req(port, "GET", "/index.html")[0] == 404
    assert _req(port, "GET", "/pointcloud/missing.ply")[0] == 400


def test_failed_click_leaves_the_session_clean_and_save_uses_the_basename(server):
    """A malformed /segment must not stay in the prompt list (every later click would carry it); /save names the file after the cloud's
    base name whatever directory it was loaded from; an oversized body is refused before it is read."""
    port, sess, pred, pts, tmp = server
    (tmp / "sub").mkdir()
    (tmp / "sub" / "deep.ply").write_bytes((tmp / "toy.ply").read_bytes())
    st, out, _ = _req(port, "GET", "/pointcloud/sub/deep.ply")
    assert st == 200
    xyz = np.array(out["xyz"]).reshape(-1, 3)
    for bad in ([0.1, 0.2], [0.1, float("nan"), 0.0], "xyz"):
        st, out, _ = _req(port, "POST", "/segment", {"prompt_point": bad, "prompt_label": 1})
        assert st == 400 and sess.prompts == [] and sess.labels == []
    pred.fail_next = True
    st, out, _ = _req(port, "POST", "/segment", {"prompt_point": xyz[2].tolist(), "prompt_label": 1})
    assert st == 400 and sess.prompts == [] and sess.labels == []
    st, out, _ = _req(port, "POST", "/segment", {"prompt_point": xyz[3].tolist(), "prompt_label": 1})
    assert st == 200 and len(sess.prompts) == 1 and pred.calls[-1][0].shape == (1, 1, 3)
    assert _req(port, "POST", "/next")[0] == 200 and _req(port, "POST", "/save")[1] == {"status": "saved"}
    assert (tmp / "results" / "deep.npy").exists()
    c = http.client.HTTPConnection("127.0.0.1", port, timeout=10)
    c.putrequest("POST", "/segment"); c.putheader("Content-Length", str(1 << 30)); c.endheaders()
    assert c.getresponse().status == 413


def test_static_routes_and_path_sanitisation(tmp_path):
    """GET /, /static/<path>, /mesh/<path> serve the front end's files like demo/app.py:71-89; every URL path stays inside its
    root directory (the reference's Flask send_static_file guarantees the same)."""
    static = tmp_path / "static"
    (static / "models" / "Rhino").mkdir(parents=True)
    (static / "index.html").write_text("<html>demo</html>")
    (static / "viewer.js").write_text("console.log(1)")
    (static / "models" / "Rhino" / "rhino.obj").write_text("v 0 0 0")
    (tmp_path / "secret.txt").write_text("top secret")
    (static / "models" / "toy.ply").write_text("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\n"
                                               "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n0 0 0 1 2 3\n")
    sess = DemoSession(FakePredictor(), models_dir=str(static / "models"), device="cpu", static_dir=str(static))
    srv = serve(sess, "127.0.0.1", 0)
    threading.Thread(target=srv.serve_forever, daemon=True).start()
    port = srv.server_address[1]
    try:
        def get(path):
            c = http.client.HTTPConnection("127.0.0.1", port, timeout=10)
            c.request("GET", path)
            r = c.getresponse()
            return r.status, r.read(), dict(r.getheaders())
        st, body, hdr = get("/")
        assert st == 200 and body == b"<html>demo</html>" and hdr["Content-Type"].startswith("text/html")
        st, body, hdr = get("/static/viewer.js")
        assert st == 200 and body == b"console.log(1)" and hdr["Content-Type"] == "application/javascript"
        st, body, _ = get("/mesh/Rhino/rhino.obj")
        assert st == 200 and body == b"v 0 0 0"
        assert get("/static/nope.js")[0] == 404
        for evil in ("/static/../secret.txt", "/static/%2e%2e/secret.txt", "/mesh/../../secret.txt", "/static//etc/passwd", "/pointcloud/../../secret.txt",
                     "/pointcloud/%2e%2e%2f%2e%2e%2fsecret.txt"):
            st, body, _ = get(evil)
            assert st in (400, 404) and b"top secret" not in body, evil
        assert get("/pointcloud/toy.ply")[0] == 200
    finally:
        srv.shutdown()
    from point_sam_amd.demo_server import safe_join
    assert safe_join(str(static), "models/toy.ply").endswith("toy.ply")
    for bad in ("../x", "a/../../x", "/abs", "models/../../secret.txt"):
        with pytest.raises(ValueError):
            safe_join(str(static), bad)
