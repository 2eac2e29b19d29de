"""One frame of Renderers.Render split over the GPUs of a node, one process per GPU.

The reference is a single-GPU program (madarch/madarch-renderers.adb:302-321 runs
its passes back to back); what follows is the build's own design (DESIGN.md
"Multi-GPU").  Per frame and rank r of N:

  0. Frame_Begin: the library picks the atlas set of this frame (frames are kept in flight)
  1. radiance pass for the probes [r P/N, (r+1) P/N)        -- no communication
  2. all-gather of the radiance atlas slices                 -- RCCL over xGMI, INSIDE the library
  3. irradiance pass -- by default for ALL probes on every rank (one workgroup per probe: it
     takes as long for 512 probes as for 64, and step 4 disappears; MDH_OPT_IRRADIANCE_ALL); it
     needs the whole radiance atlas anyway (corner-sample bleed, update_probe_irradiance.glsl:26-31)
  4. with MDH_OPT_IRRADIANCE_ALL = 0: own probes only in step 3, then all-gather of the irradiance slices
  5. volumetric passes, replicated (they depend on the camera only)
  6. screen pass for the 8x8 tiles t with t mod N == r       -- no communication

The atlases are probe-major in HBM, so a rank's slice is one contiguous byte
range and the all-gather runs in place on the atlas itself.

The exchange lives behind the C ABI: the ranks' renderers join a communicator (mdh_comm_unique_id /
mdh_comm_init, librccl opened by the library) and Renderers.Render of every rank is then the whole
schedule above -- `establish` below is all a host does.  torch.distributed appears here only as the
CONTROL plane of a run (gloo, CPU tensors: handing the 128-byte id round, agreeing on a fall-back,
barriers) and in the fall-back exchange through host memory (HostExchange) that a run takes when the
communicator cannot be formed.  DeviceExchange (the collective issued from torch on the library's
stream) is the earlier form, kept for callers that already own an RCCL process group.
"""
import ctypes as C

import numpy as np

from . import _binding as B


class HostExchange:
    """Atlas exchange through host memory and a torch.distributed group: the CPU tests of the sharded path (gloo)
    and, on GPUs, every probe count the world size does not divide (`device` = where the backend wants its tensors:
    RCCL needs them on the GPU).  Slices may be uneven or empty (more ranks than probes): every rank sends
    ceil(P / N) probes' worth, padded, and takes from rank q the first P(q+1)/N - Pq/N of them."""

    def __init__(self, dist, group=None, device=None):
        self.dist, self.group, self.device = dist, group, device

    name = "host exchange"

    def all_gather(self, renderer, tex, rank, world):
        import torch
        P = renderer.Probe_Total()
        bounds = slice_bounds(P, world)
        b, e = bounds[rank]
        cap = max(hi - lo for lo, hi in bounds)
        mine = torch.from_numpy(renderer.Read_Atlas_Slice(tex, b, e - b))
        send = torch.zeros((cap,) + tuple(mine.shape[1:]), dtype=torch.float32)
        send[:e - b] = mine
        if self.device is not None:
            send = send.to(self.device)
        outs = [torch.empty_like(send) for _ in range(world)]
        self.dist.all_gather(outs, send, group=self.group)
        for q, (lo, hi) in enumerate(bounds):
            if q != rank and hi > lo:
                renderer.Write_Atlas_Slice(tex, lo, outs[q][:hi - lo].cpu().numpy())


def slice_bounds(P, world):
    """probe slice [begin, end) of every rank: the split own_probes() of the library makes (mdh_api.hip)"""
    return [(P * r // world, P * (r + 1) // world) for r in range(world)]


def slice_bytes(P, res, texel_bytes, rank, world):
    """(offset, size) in bytes of a rank's slice of a probe-major atlas of res x res texels per probe: what
    mdh_atlas_device_ptr reports and the in-place all-gather of DeviceExchange relies on"""
    lo, hi = slice_bounds(P, world)[rank]
    per = res * res * texel_bytes
    return per * lo, per * (hi - lo)


def make_exchange(dist, renderer, device, world, group=None):
    """The exchange a sharded run uses: in place on the device when every rank's slice has the same size (RCCL's
    all-gather takes equal counts), through the host otherwise."""
    if renderer.Probe_Total() % world == 0:
        return DeviceExchange(dist, renderer, device, group)
    return HostExchange(dist, group, device)


class _DevicePtr:
    """Exposes a raw HIP allocation to torch without a copy (array interface v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False),
                                         "version": 2, "strides": None}


class DeviceExchange:
    """In-place RCCL all-gather on the probe-major atlases (backend "nccl" = RCCL):
    output = the whole atlas, input = this rank's slice of it where it lies.
    The collective is issued with the renderer's probe stream as torch's current
    stream, so it is ordered between the probe passes of the open frame and the
    screen pass of the previous frame keeps running beside it."""

    def __init__(self, dist, renderer, device, group=None):
        import torch
        self.dist, self.group, self.torch = dist, group, torch
        self.device = device
        self._streams = {}
        self._views = {}
        self._slices = {}

    def _stream(self, renderer):
        # asked per call: the library runs the probe passes on its probe stream when frames are kept in
        # flight and on its main stream when they are not (MDH_OPT_FRAME_OVERLAP, caller-supplied streams)
        b = renderer._b
        ptr = C.c_void_p()
        b.check(b.probe_stream(renderer._h, C.byref(ptr)))
        key = ptr.value or 0
        if key not in self._streams:
            self._streams[key] = self.torch.cuda.ExternalStream(key, device=self.device)
        return self._streams[key]

    def _view(self, renderer, tex):
        # (re)query: there are two atlas sets, and they are reallocated when the format option changes
        b = renderer._b
        ptr, total, off, own = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64()
        b.check(b.atlas_device_ptr(renderer._h, tex, C.byref(ptr), C.byref(total), C.byref(off), C.byref(own)))
        key = (tex, ptr.value, total.value)
        if key not in self._views:
            if len(self._views) >= 8:
                self._views.clear()
            self._views[key] = self.torch.as_tensor(_DevicePtr(ptr.value, total.value), device=self.device)
        return self._views[key], off.value, own.value, total.value

    def all_gather(self, renderer, tex, rank, world):
        full, off, own, total = self._view(renderer, tex)
        res = renderer.Probes.Radiance_Resolution if tex == B.TEX_RADIANCE else renderer.Probes.Irradiance_Resolution
        P = renderer.Probe_Total()
        if (off, own) != slice_bytes(P, res, total // (P * res * res), rank, world):
            raise RuntimeError("the library's slice of rank %d / %d (%d bytes at %d) is not the one the exchange assumes" % (rank, world, own, off))
        if own * world != total:
            raise ValueError("probe count %d is not divisible by the world size %d: use make_exchange(), which picks the host exchange" % (renderer.Probe_Total(), world))
        # in place: the rank's slice is the input where it lies in the output (sendbuff = recvbuff + rank * count,
        # the in-place form the collective defines) -- no staging copy, one kernel on the probe chain
        key = (full.data_ptr(), off, own)
        mine = self._slices.get(key)
        if mine is None:
            if len(self._slices) >= 16:
                self._slices.clear()
            mine = self._slices[key] = full[off:off + own]
        with self.torch.cuda.stream(self._stream(renderer)):
            self.dist.all_gather_into_tensor(full, mine, group=self.group)


def establish(renderer, rank, world, dist, group=None, trial_frames=2, timeout_s=120.0, log=None, backends=("rccl", "peer", "host")):
    """Form the exchange of a `world`-rank run INSIDE the library and prove it on a few frames; fall back -- in this same
    process, on every rank together -- to the next backend when that fails anywhere:

      "rccl"  the communicator (mdh_comm_init): in-place all-gather on the probe stream;
      "peer"  the peer exchange (mdh_peer_init): device-to-device copies between the processes of one node -- no
              collective library, and the one device-resident form that several ranks can run on ONE GPU;
      "host"  the slices through host memory over the control plane's group (HostExchange).

    `dist` is the control plane: a torch.distributed module whose `group` works on CPU tensors (gloo).  Returns
    (exchange, how): (None, "rccl") or (None, "peer") when Renderers.Render now carries the exchange itself, or
    (HostExchange, reason).  A rank whose exchange never returns is cut loose by a watchdog (Comm_Abort, after `timeout_s`)
    instead of hanging the run; a rank whose renderer stays stuck even then raises.  Whatever the trial frames left
    behind is wiped on every path: frame k of the run is the same frame whichever backend carries it."""
    import threading

    import torch

    def agree(ok):  # True only if every rank says so
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(t.item())

    say = log or (lambda msg: None)
    if world == 1:
        return None, "single rank"

    def wipe():  # the trial's frames must leave nothing behind (on success as well as on failure)
        try:
            renderer.Finish()
            for tex in (B.TEX_RADIANCE, B.TEX_IRRADIANCE):
                renderer.Write_Texture(tex, np.zeros(renderer.Texture_Shape(tex), dtype=np.float32))
            if hasattr(renderer, "Reset_Pass_Times"):
                renderer.Reset_Pass_Times()
        except B.MadarchError:
            pass

    def trial(barrier):
        """`trial_frames` frames behind a watchdog; returns None or the reason of the failure"""
        err = []

        def run():
            try:
                for _ in range(trial_frames):
                    renderer.Render()
                barrier()
            except B.MadarchError as e:
                err.append(str(e))

        th = threading.Thread(target=run, daemon=True)
        th.start()
        th.join(timeout_s)
        if th.is_alive():
            say("rank %d: the trial frames did not return within %.0f s: aborting the exchange" % (rank, timeout_s))
            renderer.Comm_Abort()
            th.join(30.0)
            if th.is_alive():
                raise RuntimeError("rank %d: the renderer is stuck in an exchange that the abort did not release" % rank)
            err.append("an exchange did not return within %.0f s" % timeout_s)
        return err[0] if err else None

    reasons = []
    has_comm = hasattr(renderer._b, "comm_init")
    tried_frames = False
    if "rccl" in backends:
        reason = None
        # 1. the id: 128 bytes from rank 0 (all zeros = rank 0 could not make one).  The other ranks only ask whether
        # they can load librccl at all (a rank that cannot must say so BEFORE the others enter the collective join) --
        # ncclGetUniqueId on every rank would leave a bootstrap listener behind on each
        ident = torch.zeros(B.COMM_ID_BYTES, dtype=torch.uint8)
        ok = has_comm
        if has_comm:
            try:
                if rank == 0 or not hasattr(renderer, "Comm_Available"):
                    mine = renderer.Comm_Unique_Id()
                    if rank == 0:
                        ident = torch.frombuffer(bytearray(mine), dtype=torch.uint8).clone()
                else:
                    renderer.Comm_Available()
            except B.MadarchError as e:
                ok, reason = False, str(e)
        else:
            reason = "the engine has no communicator"
        dist.broadcast(ident, src=0, group=group)
        # 2. join (collective inside RCCL: only entered when every rank is going to)
        if agree(ok):
            joined = []

            def join():
                try:
                    renderer.Comm_Init(ident.numpy().tobytes(), rank, world)
                    joined.append(None)
                except B.MadarchError as e:
                    joined.append(str(e))

            th = threading.Thread(target=join, daemon=True)
            th.start()
            th.join(timeout_s)
            if th.is_alive():
                raise RuntimeError("rank %d: ncclCommInitRank did not return within %.0f s" % (rank, timeout_s))
            if joined[0] is not None:
                ok, reason = False, joined[0]
            if not agree(ok):
                if ok:
                    renderer.Comm_Abort()
                ok, reason = False, reason or "a peer could not join the communicator"
        else:
            ok, reason = False, reason or "a peer has no communicator"
        # 3. trial frames behind a watchdog
        if ok:
            tried_frames = True
            reason = trial(renderer.Comm_Barrier)
            if reason is not None:
                ok = False
                try:
                    renderer.Comm_Abort()
                except B.MadarchError:
                    pass
            if not agree(ok):
                if ok:
                    renderer.Comm_Abort()
                ok, reason = False, reason or "a peer's trial frames failed"
        if ok:
            wipe()
            return None, "rccl"
        reasons.append("rccl: " + (reason or "a peer fell back"))
        say("rank %d: no RCCL communicator: %s" % (rank, reason))
    if "peer" in backends:
        reason = None
        ok = hasattr(renderer._b, "peer_init")
        blob = torch.zeros(B.PEER_BLOB_BYTES, dtype=torch.uint8)
        if ok:
            try:
                if tried_frames:
                    wipe()
                blob = torch.frombuffer(bytearray(renderer.Peer_Export()), dtype=torch.uint8).clone()
            except B.MadarchError as e:
                ok, reason = False, str(e)
        else:
            reason = "the engine has no peer exchange"
        blobs = [torch.zeros(B.PEER_BLOB_BYTES, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(blobs, blob, group=group)
        if agree(ok):
            try:
                renderer.Peer_Init(b"".join(bytes(t.numpy().tobytes()) for t in blobs), rank, world)
            except B.MadarchError as e:
                ok, reason = False, str(e)
            if not agree(ok):
                if ok:
                    renderer.Comm_Destroy()
                ok, reason = False, reason or "a peer could not open the handles"
        else:
            ok, reason = False, reason or "a peer has no peer exchange"
        if ok:
            tried_frames = True
            reason = trial(renderer.Finish)  # (no control-plane barrier in there: a rank whose frame failed would leave the others in it; agree () below is the barrier)
            if reason is not None:
                ok = False
            if not agree(ok):
                ok, reason = False, reason or "a peer's trial frames failed"
            if not ok:
                try:
                    renderer.Finish()
                    renderer.Comm_Destroy()
                except B.MadarchError:
                    pass
        if ok:
            wipe()
            dist.barrier(group=group)  # (nobody starts the run's frames against atlases a peer is still wiping)
            return None, "peer"
        reasons.append("peer: " + (reason or "a peer fell back"))
        say("rank %d: no peer exchange: %s" % (rank, reason))
    # the fall-back: slices through host memory over the control plane's group
    if tried_frames:  # whatever the trial left in the atlases differs from rank to rank: start from the empty state again
        wipe()
    renderer.Set_Option(B.OPT_WORLD, world)
    renderer.Set_Option(B.OPT_RANK, rank)
    why = "; ".join(reasons) or "asked for"
    say("rank %d: falling back to the host exchange: %s" % (rank, why))
    return HostExchange(dist, group), why


class ShardedFrame:
    """Drives one renderer per rank through the six steps above."""

    def __init__(self, renderer, rank, world, exchange):
        self.R, self.rank, self.world, self.exchange = renderer, rank, world, exchange
        if (renderer.Get_Option(B.OPT_RANK), renderer.Get_Option(B.OPT_WORLD)) != (rank, world):  # (a communicator has set them already)
            renderer.Set_Option(B.OPT_WORLD, world)
            renderer.Set_Option(B.OPT_RANK, rank)

    def Render(self):
        R = self.R
        if self.exchange is None:  # the library's own frame: with a communicator (Comm_Init) it runs the exchanges itself
            R.Render()
            return
        # the same frame with the exchanges between its probe passes; the library keeps such frames
        # in flight like its own (MDH_OPT_FRAME_OVERLAP): probe passes and collectives on the probe
        # stream, the screen pass of the previous frame beside them
        R.Frame_Begin()
        if R.Get_Option(B.OPT_SCREEN_MODE) == 0:
            R.Frame_Probe_Pass(B.PASS_RADIANCE)
            if self.exchange is not None:
                self.exchange.all_gather(R, B.TEX_RADIANCE, self.rank, self.world)
            R.Frame_Probe_Pass(B.PASS_IRRADIANCE)
            # by default every rank updates every probe's irradiance from the gathered radiance
            # (MDH_OPT_IRRADIANCE_ALL): the pass takes the same time and this exchange disappears
            if self.exchange is not None and not R.Get_Option(B.OPT_IRRADIANCE_ALL):
                self.exchange.all_gather(R, B.TEX_IRRADIANCE, self.rank, self.world)
        R.Frame_End()

    def Gather_Framebuffer(self, dist=None, group=None):
        """Sum of the ranks' framebuffers (each pixel is non-zero on one rank only);
        not part of the timed frame."""
        img = self.R.Read_Framebuffer()
        if self.world == 1 or dist is None:
            return img
        import torch
        t = torch.from_numpy(np.ascontiguousarray(img))
        dist.all_reduce(t, group=group)
        return t.numpy()
