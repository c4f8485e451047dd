"""RangeNormalizer -- same contract as the reference's data/dataset.py:5-53 (px / frame index <-> [dst0, dst1]).
Tiny affine host-side glue around the kernels; unlike the reference its default device is 'cpu' so that
CPU-only plumbing (dino_tracker.py:29 constructs one before any model exists) does not need a GPU."""
import torch


class RangeNormalizer(torch.nn.Module):
    def __init__(self, shapes: tuple, device="cpu"):
        super().__init__()
        self.register_buffer("normalizer", torch.tensor(shapes).float().to(device) - 1)

    @staticmethod
    def _cols(dims):
        """`dims` as a slice when it is a run of consecutive columns (every call site: [0, 1, 2], [0, 1], [2]) -- indexing a
        device tensor with a Python list builds an index tensor on the host and copies it over, a blocking pageable copy
        each time (they were ~30 of the training iteration's host stalls)."""
        dims = list(dims)
        if dims == list(range(dims[0], dims[0] + len(dims))):
            return slice(dims[0], dims[0] + len(dims))
        return dims

    def forward(self, x, dst=(0, 1), dims=[0, 1, 2]):
        c = self._cols(dims)
        out = x.clone()
        out[:, c] = (dst[1] - dst[0]) * (x[:, c] / self.normalizer[c]) + dst[0]
        return out

    def unnormalize(self, normalized_x: torch.Tensor, src=(0, 1), dims=[0, 1, 2]):
        c = self._cols(dims)
        x = normalized_x.clone()
        x[:, c] = ((normalized_x[:, c] - src[0]) / (src[1] - src[0])) * self.normalizer[c]
        return x


_PINNED = {}


def stage_to_device(host: torch.Tensor, device) -> torch.Tensor:
    """A small host tensor -> device without stalling the host: through a ring of pinned buffers and an asynchronous copy (a
    pageable source makes the copy wait for the stream's queued work first: the training loop's index tensors were 250 such
    waits per iteration).  On a CPU `device` this is the identity."""
    device = torch.device(device)
    if device.type != "cuda":
        return host.to(device)
    key = (host.dtype, device.index)
    ring = _PINNED.setdefault(key, {"bufs": [], "events": [], "next": 0})
    n = host.numel()
    if not ring["bufs"]:
        for _ in range(64):
            ring["bufs"].append(torch.empty(4096, dtype=host.dtype).pin_memory())
            ring["events"].append(None)
    if n > 4096:
        return host.to(device)
    i = ring["next"]
    ring["next"] = (i + 1) % len(ring["bufs"])
    if ring["events"][i] is not None:
        ring["events"][i].synchronize()     # 64 copies ago: long done
    buf = ring["bufs"][i][:n].view(host.shape) if host.is_contiguous() else None
    if buf is None:
        host = host.contiguous()
        buf = ring["bufs"][i][:n].view(host.shape)
    buf.copy_(host)
    out = buf.to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    ring["events"][i] = ev
    return out


class LongRangeSampler(torch.nn.Module):
    """Training-pair sampler of the reference (data/dataset.py:56-208): from optical-flow trajectories [N, T, 2] (NaN where
    a point is not tracked) draw `batch_size` pairs (point at t1, same point at t2), both times inside one random set of
    `num_frames` frames, `fg_traj_ratio` of them from the foreground trajectories.  Random numbers are consumed in the
    reference's order (randperm over frames, randperm over trajectories, multinomial over each row's valid frames), so a
    seeded run draws the same pairs."""

    CHUNK = 200_000  # trajectories kept on the device at a time when the full set stays on the host

    def __init__(self, batch_size, fg_trajectories=None, bg_trajectories=None, fg_traj_ratio=0.5, num_frames=None,
                 keep_in_cpu=False) -> None:
        super().__init__()
        self.batch_size, self.num_frames, self.fg_traj_ratio = batch_size, num_frames, fg_traj_ratio
        self.keep_in_cpu = keep_in_cpu
        self.max_traj_size = self.CHUNK
        self.gpu_batch_index = 0
        self._host = {}
        for name, traj in (("fg", fg_trajectories), ("bg", bg_trajectories)):
            valid, can = self.get_valid_trajectories(traj)
            if keep_in_cpu:  # the full set stays on the host, one chunk on the device (load_next_batch rotates)
                self._host[name] = (valid, can)
                valid, can = valid[:self.CHUNK].cuda(), can[:self.CHUNK].cuda()
            setattr(self, f"{name}_valid_trajectories", valid)
            setattr(self, f"{name}_can_sample", can)
        self.vid_len = self.fg_valid_trajectories.shape[1]

    @staticmethod
    def get_valid_trajectories(trajectories):
        """Rows with at least two tracked frames, and their per-frame validity (dataset.py:100-106)."""
        can_sample = trajectories.isnan().any(dim=-1).logical_not()
        keep = can_sample.sum(dim=1) > 1
        return trajectories[keep], can_sample[keep]

    def load_next_batch(self):
        """dataset.py:108-134: rotate the device-resident chunk of a host-resident trajectory set."""
        if not self.keep_in_cpu:
            return
        self.gpu_batch_index += 1
        for name, (valid, can) in self._host.items():
            n_chunks = -(-valid.shape[0] // self.CHUNK)
            lo = (self.gpu_batch_index % n_chunks) * self.CHUNK
            setattr(self, f"{name}_valid_trajectories", valid[lo:lo + self.CHUNK].cuda())
            setattr(self, f"{name}_can_sample", can[lo:lo + self.CHUNK].cuda())
        self._can_host = {}   # the host copies the frame-set re-draw decides on (DinoTrackerSampler._draw_frame_set)

    def get_point_correspondences_for_num_frames(self, valid_trajectories, can_sample, batch_size):
        """dataset.py:167-193."""
        n, t, _ = valid_trajectories.shape
        dev = valid_trajectories.device
        while True:  # a frame set in which at least two trajectories have two tracked frames
            frame_indices = torch.randperm(t, device=dev)[:self.num_frames]
            rows = (can_sample[:, frame_indices].sum(dim=1) >= 2).nonzero()[:, 0]
            if rows.numel() >= 2:
                break
        rows = rows[torch.randperm(rows.numel(), device=dev)[:batch_size]]
        allowed = torch.zeros_like(can_sample[rows])
        allowed[:, frame_indices] = can_sample[rows][:, frame_indices]
        t1, t2 = allowed.float().multinomial(2, replacement=False).unbind(dim=1)
        pick = lambda tt: torch.cat([valid_trajectories[rows, tt], tt[:, None].to(valid_trajectories.dtype)], dim=-1)
        return pick(t1), pick(t2)

    def get_fg_batch_size(self):
        return int(self.batch_size * self.fg_traj_ratio)

    # ---- device-side sampling (no host read of device data) -------------------------------------------------------------
    def sample_rows_on_device(self, valid_trajectories, can_sample, frame_indices, batch_size):
        """The same distribution as get_point_correspondences_for_num_frames for a GIVEN frame set, without leaving the
        device: one uniform key per trajectory, the `batch_size` largest keys among the rows that are tracked in two of
        the frames (= a uniformly random subset without replacement, what `rows[randperm(len)[:k]]` is), then two distinct
        tracked frames per row (uniformly among the row's tracked frames, as the reference's multinomial over the row's validity).  Fewer eligible rows than
        `batch_size` give rows flagged invalid (weight 0 in the losses) instead of a shorter batch -- shapes stay static.
        Returns (t1 [B, 3], t2 [B, 3], local1 [B], local2 [B] (positions in `frame_indices`), ok [B])."""
        dev = valid_trajectories.device
        can_f = can_sample[:, frame_indices]                                    # [N, F]
        eligible = can_f.sum(dim=1) >= 2
        keys = torch.where(eligible, torch.rand(can_f.shape[0], device=dev), torch.full((), -1.0, device=dev))
        k = min(batch_size, can_f.shape[0])
        from .train_ops import topk_rows      # (torch.topk, on a path that survives graph capture however many trajectories there are)
        top_v, top_i = topk_rows(keys[None], k)
        rows, ok = top_i[0], top_v[0] >= 0
        # two distinct tracked frames per row, uniformly (what multinomial(2, replacement=False) over the 0 / 1 validity row
        # draws) -- as the two largest of one uniform key per tracked frame: torch.multinomial validates its input with two host
        # reads per call
        allowed = torch.where(ok[:, None], can_f[rows], torch.ones((), dtype=torch.bool, device=dev))
        fkeys = torch.where(allowed, torch.rand(allowed.shape, device=dev), torch.full((), -1.0, device=dev))
        l1, l2 = torch.topk(fkeys, 2, dim=1).indices.unbind(dim=1)
        t1f, t2f = frame_indices[l1], frame_indices[l2]
        pick = lambda tt: torch.cat([torch.nan_to_num(valid_trajectories[rows, tt]), tt[:, None].to(valid_trajectories.dtype)], dim=-1)
        return pick(t1f), pick(t2f), l1, l2, ok

    def forward(self):
        assert self.num_frames is not None, "num_frames must be specified"
        n_fg = self.get_fg_batch_size()
        fg1, fg2 = self.get_point_correspondences_for_num_frames(self.fg_valid_trajectories, self.fg_can_sample, n_fg)
        bg1, bg2 = self.get_point_correspondences_for_num_frames(self.bg_valid_trajectories, self.bg_can_sample,
                                                                 self.batch_size - n_fg)
        return torch.cat([fg1, bg1]), torch.cat([fg2, bg2])


class DinoTrackerSampler(LongRangeSampler):
    """dataset.py:211-258: the pairs plus the batch's frame set and each point's index into it, coordinates normalised by
    `range_normalizer` to `dst_range`."""

    def __init__(self, batch_size, range_normalizer, dst_range, fg_trajectories=None, bg_trajectories=None,
                 fg_traj_ratio=0.5, num_frames=None, keep_in_cpu=False) -> None:
        super().__init__(batch_size, fg_trajectories=fg_trajectories, bg_trajectories=bg_trajectories,
                         fg_traj_ratio=fg_traj_ratio, num_frames=num_frames, keep_in_cpu=keep_in_cpu)
        self.range_normalizer, self.dst_range = range_normalizer, dst_range

    def _draw_frame_set(self, name, t, generator=None, max_tries=64):
        """A random set of num_frames frames in which at least two trajectories of the `name` set have two tracked frames (the
        reference's `while True` re-draw, dataset.py:173-179), decided on a host copy of the validity table."""
        can = getattr(self, f"{name}_can_sample")
        cached = getattr(self, "_can_host", {}).get(name)
        if cached is None or cached[0] is not can:          # (the device table changes when load_next_batch rotates the chunk)
            self._can_host = dict(getattr(self, "_can_host", {}))
            self._can_host[name] = cached = (can, can.cpu())
        can_h = cached[1]
        for _ in range(max_tries):
            frames = torch.randperm(t, generator=generator)[:self.num_frames]
            if int((can_h[:, frames].sum(dim=1) >= 2).sum()) >= 2:
                return frames
        # the reference loops forever here (dataset.py:173-179).  The batch built from this draw marks the rows of the starved set
        # invalid (`valid` of forward_device; the trainer masks them out of every loss term), so training continues -- but a set that
        # cannot supply two eligible trajectories in max_tries draws is a data problem and is said out loud, not returned silently.
        # (The number of draws is variable, so the host generator's stream differs from a fixed-count draw: documented, harmless.)
        import warnings
        warnings.warn(f"DinoTrackerSampler: no set of {self.num_frames} frames with two eligible '{name}' trajectories in "
                      f"{max_tries} draws (the reference would loop forever, data/dataset.py:173-179); the '{name}' rows of this "
                      "batch are marked invalid", RuntimeWarning, stacklevel=2)
        return frames

    def invalidate_host_tables(self):
        """Drop the host copies of {fg,bg}_can_sample (call after changing those tensors IN PLACE; a replaced tensor is noticed by
        identity, load_next_batch calls this itself)."""
        self._can_host = {}

    def forward_device(self, generator=None):
        """The batch of `forward` with the frame sets drawn on the HOST (torch's CPU generator) and everything that touches
        the trajectories on the device: no device -> host read, static shapes.  Differences from `forward`, both by
        construction: (1) `frames_set_t` is the sorted union of the two DRAWN frame sets (foreground and background
        trajectories draw theirs independently, dataset.py:173), whether or not every frame ends up used by a sampled pair
        -- `forward` returns the frames the sampled pairs actually use, a subset with the same union in all but rare
        batches.  A frame set with fewer than two eligible trajectories is redrawn, as in the reference (dataset.py:175-179) --
        on the HOST, against a host copy of the (static) per-frame validity of the trajectories, so the redraw costs no
        device read.  Extra keys: "valid" [B] bool and "frames_set_t_host" (list of int)."""
        host, union = self.draw_frame_sets(generator)
        dev = self.fg_valid_trajectories.device
        staged = stage_to_device(host, dev)                                                        # one small async copy
        frames_set_t = stage_to_device(torch.tensor(union, dtype=torch.int32), dev)
        return self.batch_from_frame_sets(staged, frames_set_t, union)

    def draw_frame_sets(self, generator=None):
        """The HOST part of forward_device: the two drawn frame sets and their positions in the sorted union, as one [4, num_frames]
        int64 host tensor (rows: foreground set, background set, foreground positions, background positions) + the union (list)."""
        assert self.num_frames is not None, "num_frames must be specified"
        t = self.vid_len
        sets = [self._draw_frame_set(name, t, generator) for name in ("fg", "bg")]               # host draws
        s0, s1 = sets[0].tolist(), sets[1].tolist()
        union = sorted(set(s0) | set(s1))
        pos = {f: i for i, f in enumerate(union)}
        return torch.tensor([s0, s1, [pos[f] for f in s0], [pos[f] for f in s1]], dtype=torch.long), union

    def batch_from_frame_sets(self, staged, frames_set_t, union=None):
        """The DEVICE part of forward_device: `staged` = draw_frame_sets' table on the device, `frames_set_t` [n] int32 the union.
        Nothing here reads the host: the call can be captured in a graph whose replays see new contents of the two tensors."""
        n_fg = self.get_fg_batch_size()
        parts = []
        for j, (name, bs) in enumerate((("fg", n_fg), ("bg", self.batch_size - n_fg))):
            p1, p2, l1, l2, ok = self.sample_rows_on_device(getattr(self, f"{name}_valid_trajectories"),
                                                            getattr(self, f"{name}_can_sample"), staged[j], bs)
            parts.append((p1, p2, staged[2 + j][l1], staged[2 + j][l2], ok))
        t1_points, t2_points, src_idx, tgt_idx, ok = (torch.cat(x) for x in zip(*parts))
        t1n = self.range_normalizer(t1_points, dst=self.dst_range)
        t2n = self.range_normalizer(t2_points, dst=self.dst_range)
        target_times = t2_points[:, 2].clone()
        t1_points[:, 2] = t1n[:, 2]
        return {
            "frames_set_t": frames_set_t,
            "frames_set_t_host": union,
            "source_frame_indices": src_idx,
            "target_frame_indices": tgt_idx,
            "t1_points_normalized": t1n,
            "t2_points_normalized": t2n,
            "t1_points": t1_points,
            "target_times": target_times,
            "valid": ok,
        }

    def forward(self):
        t1_points, t2_points = super().forward()
        frames_set_t, inverse = torch.cat((t1_points[:, 2], t2_points[:, 2])).unique(return_inverse=True)
        b = t1_points.shape[0]
        t1n = self.range_normalizer(t1_points, dst=self.dst_range)
        t2n = self.range_normalizer(t2_points, dst=self.dst_range)
        t1_points[:, 2] = t1n[:, 2]  # the sources carry their NORMALISED time (dataset.py:245), unused downstream
        return {
            "frames_set_t": frames_set_t.int(),
            "source_frame_indices": inverse[:b],
            "target_frame_indices": inverse[b:],
            "t1_points_normalized": t1n,
            "t2_points_normalized": t2n,
            "t1_points": t1_points,
            "target_times": t2_points[:, 2],
        }
