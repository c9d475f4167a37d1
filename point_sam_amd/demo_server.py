"""Back end of the reference's interactive demo on the HIP predictor.

The reference's ``demo/app.py`` is a Flask app whose routes mutate module-global state and call
``sam.set_pointcloud`` / ``sam.predict_masks`` (demo/app.py:177-206).  Here the same request/response contract is a
plain class (``DemoSession``: one method per route, JSON-shaped dicts in and out, same keys and status strings) and a
thin stdlib ``http.server`` binding -- Flask / flask_cors are not needed (nor installed in this image).  The static three.js
front end (demo/static) is UI and is not rebuilt, but its files are SERVED from ``--static-dir`` exactly as the reference does,
so the reference's front end talks to this server unchanged with no second file server.

Routes (reference line):  GET / (app.py:71-73: index.html) . GET /static/<path> (:76-78) . GET /mesh/<path> (:81-89: models/<path>)
POST /sampled_pointcloud (app.py:92-108) . GET /pointcloud/<name> (:111-141) . POST /clear (:144-150)
POST /next (:153-159) . POST /save (:162-175) . POST /segment (:177-206).
Every path taken from a URL is resolved INSIDE its root directory (no `..`, no absolute paths, no symlink escape).

    python -m point_sam_amd.demo_server --config large --ckpt model.safetensors --models-dir demo/static/models
"""
import argparse
import json
import os
import threading
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

import numpy as np
import torch

from .evaluation import load_ply

MIME = {".html": "text/html; charset=utf-8", ".js": "application/javascript", ".css": "text/css", ".json": "application/json", ".ply": "application/octet-stream",
        ".obj": "text/plain", ".mtl": "text/plain", ".png": "image/png", ".jpg": "image/jpeg", ".jpeg": "image/jpeg", ".svg": "image/svg+xml", ".ico": "image/x-icon"}


def safe_join(root: str, rel: str) -> str:
    """`rel` (a URL path, possibly percent-encoded) resolved under `root`; ValueError if it would leave the directory."""
    from urllib.parse import unquote
    rel = unquote(rel).replace("\\", "/")
    if rel.startswith("/") or "\x00" in rel:
        raise ValueError(f"illegal path {rel!r}")
    base = os.path.realpath(root)
    full = os.path.realpath(os.path.join(base, rel))
    if full != base and not full.startswith(base + os.sep):
        raise ValueError(f"path {rel!r} leaves the served directory")
    return full


class DemoSession:
    """The demo's state machine.  ``predictor`` needs ``set_pointcloud(xyz, rgb)`` and
    ``predict_masks(points, labels, prompt_mask, multimask_output) -> (mask, scores, logits)``."""

    def __init__(self, predictor, models_dir: str = ".", pointcloud: str = None, output_dir: str = "results", device="cuda", static_dir: str = None):
        self.predictor = predictor
        self.models_dir, self.pointcloud, self.output_dir = models_dir, pointcloud, output_dir
        self.static_dir = static_dir           # the front end's files (reference: demo/static); None = not served
        self.device = torch.device(device)
        self.lock = threading.Lock()          # the reference is single-threaded; requests are serialised here
        self.pc_xyz = self.pc_rgb = None
        self.obj_path = None
        self.masks = []
        self._reset_prompts()
        self.segment_mask = None

    def _reset_prompts(self):
        self.prompts, self.labels, self.prompt_mask = [], [], None

    # ---- routes -----------------------------------------------------------------------------------------------
    def sampled_pointcloud(self, data: dict) -> dict:
        """Client-side sampled cloud: {"points": {i: v}, "colors": {i: v}} flattened xyz / rgb (app.py:92-108)."""
        pts = np.array(list(data["points"].values()), dtype=np.float64).reshape(-1, 3)
        col = np.array(list(data["colors"].values()), dtype=np.float64).reshape(-1, 3)
        with self.lock:
            self.pc_xyz = torch.from_numpy(pts).to(self.device).float()[None]
            self.pc_rgb = torch.from_numpy(col).to(self.device).float()[None]
        return {"response": "success"}

    def load_pointcloud(self, path: str) -> dict:
        """Loads an ASCII PLY from models_dir, normalises it into the unit ball, rgb / 255 (app.py:111-141).  As in the
        reference, a configured --pointcloud overrides the requested name."""
        name = self.pointcloud or path
        pts = load_ply(safe_join(self.models_dir, name))
        xyz, rgb = pts[:, :3], pts[:, 3:6] / 255
        shift = xyz.mean(0)
        scale = np.linalg.norm(xyz - shift, axis=-1).max()
        xyz = (xyz - shift) / scale
        with self.lock:
            self.obj_path = name
            self.pc_xyz = torch.from_numpy(xyz).to(self.device).float()[None]
            self.pc_rgb = torch.from_numpy(rgb).to(self.device).float()[None]
        return {"xyz": xyz.flatten().tolist(), "rgb": rgb.flatten().tolist()}

    def static_file(self, rel: str):
        """(bytes, mime type) of a front-end file under static_dir (app.py:71-89)."""
        if self.static_dir is None:
            raise FileNotFoundError("no --static-dir configured: front-end files are not served")
        full = safe_join(self.static_dir, rel)
        if not os.path.isfile(full):
            raise FileNotFoundError(rel)
        with open(full, "rb") as f:
            return f.read(), MIME.get(os.path.splitext(full)[1].lower(), "application/octet-stream")

    def clear(self) -> dict:
        with self.lock:
            self._reset_prompts()
            self.segment_mask = None
        return {"status": "cleared"}

    def next(self) -> dict:
        with self.lock:
            if self.segment_mask is None:
                raise ValueError("/next before any /segment: there is no mask to keep")
            self.masks.append(self.segment_mask.cpu().numpy())
            self._reset_prompts()
        return {"status": "cleared"}

    def save(self) -> dict:
        with self.lock:
            if self.pc_xyz is None or not self.masks:
                raise ValueError("/save needs a point cloud and at least one kept mask")
            os.makedirs(self.output_dir, exist_ok=True)
            stem = os.path.splitext(os.path.basename(self.obj_path or "pointcloud"))[0] or "pointcloud"      # 'sub/x.ply' -> 'x', './x.ply' -> 'x'
            np.save(os.path.join(self.output_dir, f"{stem}.npy"),
                    {"xyz": self.pc_xyz[0].cpu().numpy(), "rgb": self.pc_rgb[0].cpu().numpy(), "mask": np.stack(self.masks)})
            self._reset_prompts()
            self.segment_mask = None
        return {"status": "saved"}

    def segment(self, data: dict) -> dict:
        """One click: append the prompt, run the decoder on the cached cloud, keep the best mask's logits as the next
        dense prompt; multimask only on the first click (app.py:177-206)."""
        with self.lock:
            if self.pc_xyz is None:
                raise ValueError("/segment before a point cloud was set")
            # the click joins the session only after the predictor accepted it: a malformed point must not poison every later click
            point = np.asarray(data["prompt_point"], dtype=np.float32)
            label = int(data["prompt_label"])
            if point.shape != (3,) or not np.isfinite(point).all():
                raise ValueError("prompt_point must be three finite numbers")
            prompts, labels = self.prompts + [point.tolist()], self.labels + [label]
            pts = torch.from_numpy(np.array(prompts, dtype=np.float32)).to(self.device).float()[None]
            lab = torch.from_numpy(np.array(labels)).to(self.device)[None]
            with torch.no_grad():
                self.predictor.set_pointcloud(self.pc_xyz, self.pc_rgb)
                mask, scores, logits = self.predictor.predict_masks(pts, lab, self.prompt_mask, self.prompt_mask is None)
            self.prompts, self.labels = prompts, labels
            best = torch.argmax(scores[0])
            self.prompt_mask = logits[0][best][None]
            self.segment_mask = mask[0][best] > 0
            return {"seg": self.segment_mask.cpu().numpy().tolist()}


MAX_BODY_BYTES = 64 << 20      # a sampled 10^5-point cloud as JSON is a few MB


def make_handler(session: DemoSession, allow_origin: str = "*"):
    class Handler(BaseHTTPRequestHandler):
        def _send(self, code, obj):
            body = json.dumps(obj).encode()
            self.send_response(code)
            self.send_header("Content-Type", "application/json")
            self.send_header("Content-Length", str(len(body)))
            self.send_header("Access-Control-Allow-Origin", allow_origin)      # what flask_cors provides in the reference
            self.send_header("Access-Control-Allow-Headers", "Content-Type, Access-Control-Allow-Origin")
            self.end_headers()
            self.wfile.write(body)

        def _run(self, fn, *a):
            try:
                self._send(200, fn(*a))
            except (ValueError, KeyError, AssertionError, FileNotFoundError) as e:
                self._send(400, {"error": f"{type(e).__name__}: {e}"})

        def do_OPTIONS(self):
            self._send(200, {})

        def _send_file(self, rel):
            try:
                body, mime = session.static_file(rel)
            except ValueError as e:
                return self._send(400, {"error": f"ValueError: {e}"})
            except FileNotFoundError as e:
                return self._send(404, {"error": f"not found: {e}"})
            self.send_response(200)
            self.send_header("Content-Type", mime)
            self.send_header("Content-Length", str(len(body)))
            self.send_header("Access-Control-Allow-Origin", allow_origin)
            self.end_headers()
            self.wfile.write(body)

        def do_GET(self):
            path = self.path.split("?", 1)[0]
            if path.startswith("/pointcloud/"):
                self._run(session.load_pointcloud, path[len("/pointcloud/"):])
            elif path == "/":
                self._send_file("index.html")                                   # app.py:71-73
            elif path.startswith("/static/"):
                self._send_file(path[len("/static/"):])                         # app.py:76-78
            elif path.startswith("/mesh/"):
                self._send_file("models/" + path[len("/mesh/"):])               # app.py:81-89
            else:
                self._send(404, {"error": f"no route {path}"})

        def do_POST(self):
            try:
                n = int(self.headers.get("Content-Length") or 0)
            except ValueError:
                return self._send(400, {"error": "bad Content-Length"})
            if n < 0 or n > MAX_BODY_BYTES:
                return self._send(413, {"error": f"request body over {MAX_BODY_BYTES} bytes"})
            try:
                data = json.loads(self.rfile.read(n) or b"{}")
            except json.JSONDecodeError as e:
                return self._send(400, {"error": f"bad JSON: {e}"})
            routes = {"/sampled_pointcloud": lambda: session.sampled_pointcloud(data), "/segment": lambda: session.segment(data),
                      "/clear": session.clear, "/next": session.next, "/save": session.save}
            fn = routes.get(self.path)
            if fn is None:
                return self._send(404, {"error": f"no route {self.path}"})
            self._run(fn)

        def log_message(self, fmt, *args):  # quiet
            pass

    return Handler


def serve(session: DemoSession, host="localhost", port=5000) -> ThreadingHTTPServer:
    return ThreadingHTTPServer((host, port), make_handler(session))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", default="localhost")
    ap.add_argument("--port", type=int, default=5000)
    ap.add_argument("--config", default="large")
    ap.add_argument("--ckpt", "--ckpt_path", dest="ckpt", default=None, help="safetensors checkpoint (random weights if omitted)")
    ap.add_argument("--pointcloud", default=None)
    ap.add_argument("--models-dir", default="demo/static/models")
    ap.add_argument("--static-dir", default="demo/static", help="the reference front end's files (index.html, *.js, models/)")
    ap.add_argument("--precision", default="f16x3")
    args = ap.parse_args()
    from .predictor import PointSAMPredictor
    pred = PointSAMPredictor.from_config(args.config, args.ckpt, precision=args.precision)
    srv = serve(DemoSession(pred, args.models_dir, args.pointcloud, static_dir=args.static_dir), args.host, args.port)
    print(f"Point-SAM demo back end on http://{args.host}:{args.port}")
    srv.serve_forever()


if __name__ == "__main__":
    main()
