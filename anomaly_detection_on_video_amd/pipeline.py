"""extract -> MIL score stream: the whole hot path resident on the GPU.

A stream of crop-clips in the reference's order (video-major, clip, crop:
`(n_clips, 10, 2048)` per video, extract_features.py:93-100) is cut into global batches; every rank
runs the I3D backbone on its contiguous block, the 2048-d rows are all-gathered (RCCL), written
into a per-video ring and, whenever a video's last crop-clip has arrived, the rank that owns the
video scores it: add_magnitude (dataset.py:121-124) -> MGFN eval forward (runner.py:42-50) ->
clip-level anomaly scores.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import List, Optional, Sequence, Tuple, Union

import torch

from . import dist as adist
from . import mil_ops


class FrameCrops:
    """A step's input given as resized uint8 frames instead of fp32 crop-clips: crop-clips [first, first + count) of the
    video `frames` (F, FH, FW, 3) on the device, row = clip * 10 + crop (TenCrop order).  The backbone's stem reads the pixels
    itself (I3Res50.forward_frames); a `prepare` callable of step_async may return one of these."""

    def __init__(self, frames: torch.Tensor, first: int, count: int, frames_per_clip: int = 16, crop: int = 224):
        self.frames, self.first, self.count, self.frames_per_clip, self.crop = frames, first, count, frames_per_clip, crop

    def key(self):
        return ("u8", self.count, tuple(self.frames.shape[1:3]), self.frames_per_clip, self.crop)


class ExtractScoreStream:
    """`clips_per_video`: one clip count for every video, or a sequence -- video v has `clips_per_video[v % len]` clips of `ncrops`
    crop-clips each: the variable-length stream extract_features.py produces (one (n_clips, 10, 2048) array per video,
    /root/reference/extract_features.py:93-100, 104-110; UCF-Crime: 50..500 clips).  A global batch may end one video and begin the
    next; every video is scored with T = its own clip count (runner.py:42-50)."""

    GRAPH_CACHE_MAX = 8      # captured scoring passes kept (ADV_SCORE_GRAPH=1), per distinct video length
    SCORED_LOG_MAX = 4096    # entries of scored_log kept

    def __init__(self, backbone, scorer, clips_per_video: Union[int, Sequence[int]] = 32, ncrops: int = 10, local_batch: int = 32,
                 world: int = 1, rank: int = 0, feat_dim: int = 2048):
        self.backbone, self.scorer = backbone, scorer
        self._clips = [int(clips_per_video)] if isinstance(clips_per_video, int) else [int(c) for c in clips_per_video]
        if not self._clips or min(self._clips) < 1:
            raise ValueError("clips_per_video: a positive clip count, or a non-empty sequence of them")
        self.clips_per_video = self._clips[0] if len(self._clips) == 1 else tuple(self._clips)
        self.ncrops = ncrops
        self.local_batch, self.world, self.rank = local_batch, world, rank
        self.global_batch = local_batch * world
        # The ring holds the last `ring_rows` feature rows of the stream, ring_rows >= longest video + one global batch and a multiple of
        # the global batch (a batch never wraps).  Rows [0, max_video_rows) are mirrored behind the ring's end, so that every video
        # -- also one whose rows wrap around -- is ONE contiguous (n_clips * ncrops, C) window: no gather, no concatenation.
        self.max_video_rows = max(self._clips) * ncrops
        gb = self.global_batch
        self.ring_rows = -(-(self.max_video_rows + gb) // gb) * gb
        dev = next(backbone.parameters()).device
        self.ring = torch.zeros((self.ring_rows + self.max_video_rows, feat_dim), device=dev, dtype=torch.float32)
        self._vid = 0        # the oldest video whose last crop-clip has not arrived yet ...
        self._vid_start = 0  # ... and the stream position of its first row
        self.pos = 0  # global stream position (crop-clips consumed so far)
        self.videos_scored = 0
        # (video, its clip count) of the last SCORED_LOG_MAX videos this rank scored (a bounded record: the stream may run for days)
        self.scored_log: List[Tuple[int, int]] = []
        self.last_scores: Optional[torch.Tensor] = None
        # eval scoring of one video is ~200 tiny launches; ADV_SCORE_GRAPH=1 replays it as one hipGraph
        # (opt-in: measured neutral on throughput, the launches already overlap the backbone)
        self.use_graph = os.environ.get("ADV_SCORE_GRAPH") == "1"
        # one captured pass per video SHAPE, least recently used first; at most GRAPH_CACHE_MAX of them are kept (each holds its
        # static input / output and a private activation pool: a variable-length stream would otherwise grow without bound)
        self._graphs: "OrderedDict[Tuple[int, int, int], Tuple[torch.cuda.CUDAGraph, torch.Tensor, torch.Tensor]]" = OrderedDict()
        # step_async: consecutive steps alternate between `lanes` HIP streams (see step_async)
        # (measured at B=32 on one MI355X: 1 lane + batch split over 2 streams 10.12 ms/step, 2 lanes 9.96, 3 lanes 9.65,
        #  4 lanes 10.06, 6 lanes 9.89; lanes x an in-step batch split is slower than lanes alone)
        self.lanes = int(os.environ.get("ADV_PIPELINE_LANES", "3"))
        self._lane_streams: List[torch.cuda.Stream] = []
        self._lane_next = 0
        self._ordered: Optional[torch.cuda.Event] = None  # the previous step's gather + ring update (+ its scoring's ring reads)
        self._scored_done: Optional[torch.cuda.Event] = None  # end of the last scoring
        self._last_done: Optional[torch.cuda.Event] = None    # end of the last step issued
        self._after_ring_read = None                           # step_async: hook run once a scoring has issued its ring reads
        self._table_dims: set = set()                      # (batch, T, H, W) whose lazily built operands exist
        self._tables_built: Optional[torch.cuda.Event] = None

    @torch.no_grad()
    def step(self, local_clips: torch.Tensor) -> Tuple[torch.Tensor, List[Tuple[int, torch.Tensor]]]:
        """local_clips: this rank's (local_batch, 3, T, H, W) block of the next global batch.
        Returns (gathered features (global_batch, 2048), [(video index, scores (clips,)) ...])."""
        feats = self.backbone(local_clips).reshape(local_clips.shape[0], -1)
        gathered = adist.all_gather_rows(feats) if self.world > 1 else feats
        return gathered, self.ingest(gathered)

    @torch.no_grad()
    def step_async(self, local_clips: torch.Tensor, prepare=None) -> "StepHandle":
        """`step` without the implied ordering against the caller's stream: step i runs on lane i % lanes (a HIP
        stream of its own: backbone, all-gather, ring ingest, scoring), so the backbone of step i+1 runs beside the
        tail, the gather and the scoring of step i and memory-bound launches of one step overlap MFMA-bound ones of the
        other.  Only the ring update (gather + ingest + scoring) is ordered step after step, by an event.  The
        returned handle's `result()` makes the caller's current stream wait for the step.

        `prepare`: optional callable run on the lane before the backbone, e.g. the H2D copy of pinned host pixels
        and the uint8 -> fp32 normalise/permute pass, so that PCIe transfers of one step overlap the other lanes'
        compute (`lambda host_u8: mil_ops.normalize_permute_u8(host_u8.to(dev, non_blocking=True))`)."""
        dev = self.ring.device
        if self.lanes <= 1:
            x = local_clips if prepare is None else prepare(local_clips)
            if isinstance(x, FrameCrops):
                raise ValueError("FrameCrops input needs the lane form of step_async (lanes > 1)")
            g, s = self.step(x)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            return StepHandle(ev, g, s)
        while len(self._lane_streams) < self.lanes:
            self._lane_streams.append(torch.cuda.Stream(device=dev))
        lane = self._lane_streams[self._lane_next]
        self._lane_next = (self._lane_next + 1) % self.lanes
        # packed weights and gather tables are created lazily on the stream that first needs them: create them
        # before the lanes fork (dims known), or on this lane with an event every later lane step waits for
        tables = getattr(self.backbone, "ensure_tables", None)
        def table_key(t):
            return (t.shape[0],) + tuple(t.shape[2:])  # the resolved kernel choices depend on the batch size too

        if prepare is None and tables is not None and table_key(local_clips) not in self._table_dims:
            tables(tuple(local_clips.shape[2:]), local_clips.shape[0])
            self._table_dims.add(table_key(local_clips))
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        with torch.cuda.stream(lane):
            lane.wait_event(ready)
            if self._tables_built is not None:
                lane.wait_event(self._tables_built)
            if prepare is not None:
                local_clips = prepare(local_clips)
                fc = local_clips if isinstance(local_clips, FrameCrops) else None
                key = fc.key() if fc is not None else table_key(local_clips)
                if tables is not None and key not in self._table_dims:
                    if fc is not None:
                        self.backbone.ensure_frame_tables(tuple(fc.frames.shape[1:3]), fc.frames_per_clip, fc.crop, fc.count)
                    else:
                        tables(tuple(local_clips.shape[2:]), local_clips.shape[0])
                    self._table_dims.add(key)
                    self._tables_built = torch.cuda.Event()
                    self._tables_built.record(lane)
            else:
                local_clips.record_stream(lane)
            inner = getattr(self.backbone, "streams", 1)
            try:
                self.backbone.streams = 1  # whole-batch launches per lane; the overlap comes from the other lanes
                if isinstance(local_clips, FrameCrops):
                    fc = local_clips
                    feats = self.backbone.forward_frames(fc.frames, fc.first, fc.count, fc.frames_per_clip, fc.crop).reshape(fc.count, -1)
                else:
                    feats = self.backbone(local_clips).reshape(local_clips.shape[0], -1)
            finally:
                self.backbone.streams = inner
            if self._ordered is not None:
                lane.wait_event(self._ordered)
            gathered = adist.all_gather_rows(feats) if self.world > 1 else feats

            # The ring update is what has to run step after step.  A video's scoring (2.5 ms of small launches every
            # clips_per_video * ncrops / global_batch steps) only READS its ring rows at its very start (add_magnitude /
            # the copy into the graph's input): the next step may touch the ring as soon as that read is issued, so the
            # `_ordered` event is recorded there and the scorer runs beside the other lanes' steps instead of holding
            # them up (with the event at the end of the scoring every lane queued behind it: -2 % clips/s).
            fired = []

            def ring_read():  # (called when the LAST video of this batch has issued its ring reads)
                self._ordered = torch.cuda.Event()
                self._ordered.record(lane)
                fired.append(True)
                if self._scored_done is not None:  # scorings themselves stay ordered (lazily built operand caches)
                    lane.wait_event(self._scored_done)

            self._after_ring_read = ring_read
            if self._scored_done is not None and len(self._videos_completing()) > 1:
                # ring_read orders only the LAST video of a step behind the previous scoring; when a step completes several
                # videos of this rank the earlier ones must wait for it too (lazily built operand caches are shared)
                lane.wait_event(self._scored_done)
            try:
                scored = self.ingest(gathered)
            finally:
                self._after_ring_read = None
            done = torch.cuda.Event()
            done.record(lane)
            if scored:
                self._scored_done = done
            if not fired:
                self._ordered = done
        self._last_done = done
        return StepHandle(done, gathered, scored)

    def drain(self) -> None:
        """Make the caller's current stream wait for every step issued by step_async."""
        cur = torch.cuda.current_stream(self.ring.device)
        for ev in (self._ordered, self._scored_done, self._last_done):
            if ev is not None:
                cur.wait_event(ev)

    def video_rows(self, v: int) -> int:
        return self._clips[v % len(self._clips)] * self.ncrops

    def seek(self, pos: int) -> None:
        """Continue a stream at crop-clip `pos` (a multiple of the global batch): the videos that end before `pos` count as done,
        the one `pos` falls into is scored when its last crop-clip arrives -- from the rows ingested from here on (its earlier rows
        are whatever the ring holds: zeros on a fresh stream)."""
        if pos % self.global_batch or pos < 0:
            raise ValueError(f"seek: {pos} is not a multiple of the global batch {self.global_batch}")
        v, start = 0, 0
        while start + self.video_rows(v) <= pos:
            start += self.video_rows(v)
            v += 1
        self.pos, self._vid, self._vid_start = pos, v, start

    def _completing(self) -> List[Tuple[int, int, int]]:
        """(video, stream position of its first row, rows) of every video whose last crop-clip arrives with the NEXT global batch."""
        out, v, start = [], self._vid, self._vid_start
        while start + self.video_rows(v) <= self.pos + self.global_batch:
            out.append((v, start, self.video_rows(v)))
            start += self.video_rows(v)
            v += 1
        return out

    def _videos_completing(self) -> List[int]:
        """Videos of this rank whose last crop-clip arrives with the NEXT global batch."""
        return [v for v, _s, _n in self._completing() if v % self.world == self.rank]

    @torch.no_grad()
    def ingest(self, gathered: torch.Tensor) -> List[Tuple[int, torch.Tensor]]:
        """Append one global batch of feature rows (stream order) to the ring and score every video
        owned by this rank (video v -> rank v % world) whose last crop-clip just arrived."""
        gb = self.global_batch
        if gathered.shape[0] != gb:
            raise ValueError(f"expected {gb} rows per global batch, got {gathered.shape[0]}")
        start = self.pos % self.ring_rows
        self.ring[start : start + gb].copy_(gathered)
        if start < self.max_video_rows:  # the mirror of the ring's head
            m = min(gb, self.max_video_rows - start)
            self.ring[self.ring_rows + start : self.ring_rows + start + m].copy_(gathered[:m])
        done = self._completing()
        self.pos += gb
        if done:
            v, s, n = done[-1]
            self._vid, self._vid_start = v + 1, s + n
        mine = [(v, s, n) for v, s, n in done if v % self.world == self.rank]
        scored = []
        hook, self._after_ring_read = self._after_ring_read, None
        for v, s, n in mine:
            r0 = s % self.ring_rows
            vid = self.ring[r0 : r0 + n].view(n // self.ncrops, self.ncrops, -1)
            if v == mine[-1][0]:
                self._after_ring_read = hook  # the ring is free for the next step once the last video has been read
            scored.append((v, self.score_video(vid)))
            self.scored_log.append((v, n // self.ncrops))
            if len(self.scored_log) > self.SCORED_LOG_MAX:
                del self.scored_log[: len(self.scored_log) - self.SCORED_LOG_MAX]
        return scored

    @torch.no_grad()
    def score_video(self, feats: torch.Tensor) -> torch.Tensor:
        """(n_clips, ncrops, 2048) -> (n_clips,) anomaly scores; validation_step semantics
        (runner.py:42-50): add magnitude channel, (1, T, 10, 2049) -> (1, 10, T, 2049), eval forward."""
        self.videos_scored += 1
        if self.use_graph and feats.is_cuda and not self.scorer.training:
            # a captured graph holds raw pointers to the cached packed weights: any change of the parameters (optimizer step,
            # load_state_dict) must force a re-capture, or a replay reads stale / freed operands
            from . import mgfn_ops

            stamp = (mgfn_ops._EPOCH,) + tuple((p.data_ptr(), p._version) for p in self.scorer.parameters())
            if stamp != getattr(self, "_graphs_stamp", None):
                self._graphs.clear()
                self._graphs_stamp = stamp
            key = tuple(feats.shape)
            entry = self._graphs.get(key)
            if entry is None:
                static_in = feats.clone()
                self._score_eager(static_in)  # warm-up outside capture (lazily built operands: gather tables, packed / folded weights)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = self._score_eager(static_in)
                entry = self._graphs[key] = (graph, static_in, static_out)
                while len(self._graphs) > self.GRAPH_CACHE_MAX:
                    self._graphs.popitem(last=False)
            self._graphs.move_to_end(key)
            graph, static_in, static_out = entry
            static_in.copy_(feats)
            self._ring_read_issued()
            graph.replay()
            self.last_scores = static_out.clone()
        else:
            self.last_scores = self._score_eager(feats, self._ring_read_issued)
        return self.last_scores

    def _ring_read_issued(self) -> None:
        if self._after_ring_read is not None:
            hook, self._after_ring_read = self._after_ring_read, None
            hook()

    def _score_eager(self, feats: torch.Tensor, after_read=None) -> torch.Tensor:
        x = mil_ops.add_magnitude(feats)  # (T, 10, 2049): the only read of `feats` (ring rows)
        if after_read is not None:
            after_read()
        video = x.unsqueeze(0).permute(0, 2, 1, 3).contiguous()
        return self.scorer(video=video).scores.reshape(-1)


class HostFeeder:
    """Pinned host buffers -> the device on a copy stream of its own, `depth` device buffers deep: the H2D transfer of step
    k + 1 runs while step k computes instead of sitting at the head of its lane's chain.  `feed(host)` returns a `prepare`
    callable for `ExtractScoreStream.step_async` (it makes the lane wait for the copy and hands over the device buffer);
    `done(handle)` tells the feeder which step last read the buffer, so that the copy that re-uses it waits for that step."""

    def __init__(self, device, depth: int = 4):
        self.device, self.depth = torch.device(device), depth
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.bufs: List[Optional[torch.Tensor]] = [None] * depth
        self.ready: List[Optional[torch.cuda.Event]] = [None] * depth
        self.last_use: List[Optional[torch.cuda.Event]] = [None] * depth
        self.k = 0

    def feed(self, host: torch.Tensor, wrap=None):
        j = self.k % self.depth
        self.k += 1
        with torch.cuda.stream(self.copy_stream):
            if self.last_use[j] is not None:
                self.copy_stream.wait_event(self.last_use[j])  # the step that read this buffer last has finished
            if self.bufs[j] is None or self.bufs[j].shape != host.shape or self.bufs[j].dtype != host.dtype:
                self.bufs[j] = torch.empty(host.shape, device=self.device, dtype=host.dtype)
            self.bufs[j].copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.ready[j] = ev
        buf = self.bufs[j]

        def prepare(_host):
            torch.cuda.current_stream(self.device).wait_event(ev)
            return buf if wrap is None else wrap(buf)

        prepare.slot = j
        return prepare

    def done(self, prepare, handle: "StepHandle") -> None:
        self.last_use[prepare.slot] = handle._done


class StepHandle:
    """Result of ExtractScoreStream.step_async: tensors produced on a lane stream."""

    def __init__(self, done: torch.cuda.Event, gathered: torch.Tensor, scored: List[Tuple[int, torch.Tensor]]):
        self._done, self._gathered, self._scored = done, gathered, scored

    def result(self) -> Tuple[torch.Tensor, List[Tuple[int, torch.Tensor]]]:
        """(gathered features, [(video, scores)]) after making the current stream wait for the step."""
        cur = torch.cuda.current_stream(self._gathered.device)
        cur.wait_event(self._done)
        self._gathered.record_stream(cur)
        for _v, sc in self._scored:
            sc.record_stream(cur)
        return self._gathered, self._scored
