"""VQModel — the conditional VQGAN of SGAM on the MI355X HIP backend.

Drop-in for ``sgam.generative_sensing_module.model.VQModel`` (reference model.py:18-269) on the
inference path: same constructor signature, attribute names, ``state_dict`` keys (345 hot-path tensors;
``loss.*`` / ``perceptual_loss.*`` keys of a checkpoint are tolerated with strict=False), same
``encode`` / ``decode`` / ``forward`` / ``get_x`` signatures and return structures.  It is a plain
``nn.Module`` (no Lightning at inference): ``global_step`` = 0, ``device`` property provided.

Everything between the NCHW boundary tensors runs in libsgam_hip.so, NHWC fp32, without leaving the GPU.
The training half of the reference class (training_step / configure_optimizers / online k-means,
model.py:271-472) delegates to ``sgam_neurips22_amd.training`` (hand-written backward in
csrc/train.hip): ``training_step`` / ``configure_optimizers`` below keep the reference's call surface.
"""
import torch
import torch.nn as nn

from .. import ops
from ..point_rendering.warp import splat_to_model_input
from .modules.diffusionmodules.model import Conv2d, Decoder, Encoder
from .modules.vqvae.quantize import VectorQuantizer2 as VectorQuantizer


class VQModel(nn.Module):
    global_step = 0
    global_rank = 0

    def __init__(self, ddconfig, data_config, lossconfig, n_embed, embed_dim, phase=None, ckpt_path=None,
                 ignore_keys=['loss.discriminator'], image_key="image", colorize_nlabels=None, logdir=None,
                 use_extrapolation_mask=True, vq_step_threshold=0, monitor=None, remap=None, sane_index_shape=False,
                 online_kmeans_config=None, batch_size=None, depth_range=None):
        super().__init__()
        online_kmeans_config = online_kmeans_config or {}
        self.phase = phase
        self.online_kmeans_config = online_kmeans_config
        self.data_config = data_config
        self.logdir = logdir
        self.depth_range = depth_range
        self.n_embed = n_embed
        self.do_online_kmeans_clustering = online_kmeans_config.get('do_online_kmeans_clustering', False)
        self.use_extrapolation_mask = use_extrapolation_mask
        self.vq_step_threshold = vq_step_threshold
        self.image_key = image_key
        self.use_rgbd_integration = False
        # arithmetic of the VQGAN body: float32 = parity path (fp32-in MFMA); bfloat16 / float16 = throughput
        # path (16-bit MFMA, fp32 accumulate).  The quantiser always runs in fp32 on the fp32 latent.
        self.compute_dtype = torch.float32
        # replay the ~215 kernel launches of one forward from a captured HIP graph (opt-in: enable_hip_graph())
        self.use_hip_graph = False
        self._graphs = {}
        if self.use_extrapolation_mask:
            self.conv_in = Conv2d(5, 4, kernel_size=1)
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        # lossconfig (LPIPS + PatchGAN) is a training-only component: accepted, not instantiated.
        self.lossconfig = lossconfig
        self.quantize = VectorQuantizer(n_embed, embed_dim, beta=0.25, remap=remap, sane_index_shape=sane_index_shape,
                                        kmean_init_codebook_path=online_kmeans_config.get('kmean_init_codebook_path'))
        self.quant_conv = Conv2d(ddconfig["z_channels"], embed_dim, 1)
        self.post_quant_conv = Conv2d(embed_dim, ddconfig["z_channels"], 1)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)
        if monitor is not None:
            self.monitor = monitor

    # ---- Lightning-free conveniences the callers rely on ----
    @property
    def device(self):
        return next(self.parameters()).device

    def use_vq(self):
        return self.global_step >= self.vq_step_threshold

    def init_from_ckpt(self, path, ignore_keys=['loss'], only_keep_keys=[]):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        for k in list(sd.keys()):
            if any(ik not in k for ik in only_keep_keys):
                del sd[k]
        self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}")

    def set_compute_dtype(self, dtype):
        """'f32' (default, parity), 'bf16' or 'fp16' (throughput).  Returns self."""
        self.compute_dtype = ops.DTYPES[dtype] if isinstance(dtype, str) else dtype
        return self

    def enable_hip_graph(self, on=True):
        """Capture `forward` once per (input shape, flags, dtype) into a HIP graph (hipStreamBeginCapture through
        torch.cuda.graph — the kernels are this library's, launched on the capturing stream) and replay it on later
        calls: one graph launch instead of ~215 kernel launches, no per-launch host work, intermediates in a
        graph-private pool.  Outputs of a replay are the graph's static tensors: they are overwritten by the next
        call with the same signature (the scene loop consumes them before that).  Weights must not change while
        graphs exist (call enable_hip_graph(False) first)."""
        self.use_hip_graph = bool(on)
        if not on:
            self._graphs = {}
        return self

    def eager(self):
        """context manager: launch every kernel of `forward` eagerly even when graphs are enabled (profiling passes)"""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            old, self._graph_bypass = getattr(self, "_graph_bypass", False), True
            try:
                yield self
            finally:
                self._graph_bypass = old
        return _cm()

    def _forward_graphed(self, input, topk, extrapolation_mask, sample_number, flags):
        # inputs that live in persistent buffers of the caller (the scene loop's warp outputs, marked `_sgam_persistent`)
        # are captured by address: the graph reads them in place and no copy is paid per replay
        inplace = getattr(input, "_sgam_persistent", False) and (
            extrapolation_mask is None or getattr(extrapolation_mask, "_sgam_persistent", False))
        if self.compute_dtype == torch.float32 and ops.F32_MODE == "split":
            ops.range_flag(input.device)       # this device's flag is the registered one before anything is captured
        key = (tuple(input.shape), None if extrapolation_mask is None else tuple(extrapolation_mask.shape), topk,
               sample_number, flags, self.compute_dtype, str(input.device), ops.RANGE_FLAG_EPOCH,
               (input.data_ptr(), None if extrapolation_mask is None else extrapolation_mask.data_ptr()) if inplace else None)
        ent = self._graphs.get(key)
        if ent is None:
            sx = input if inplace else input.detach().clone()
            sm = None if extrapolation_mask is None else (extrapolation_mask if inplace else extrapolation_mask.detach().clone())
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # eager warm-up: weight packing, codebook norms, allocator pools
                for _ in range(2):                 # (guarded: a range overflow switches to the fp32-in MFMA before capture)
                    self._forward_guarded(sx, topk, sm, sample_number, flags)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._forward_eager(sx, topk, sm, sample_number, *flags)
            ent = self._graphs[key] = (graph, sx, sm, out)
        graph, sx, sm, out = ent
        if not inplace:
            sx.copy_(input)
            if sm is not None:
                sm.copy_(extrapolation_mask)
        graph.replay()
        return out

    # ---- NHWC core ----
    def _encode_nhwc(self, x, extrapolation_mask):
        """x (B,4,H,W) NCHW + mask -> pre-quant latent (B,h,w,D) NHWC fp32."""
        dt = self.compute_dtype
        if self.use_extrapolation_mask:
            h = ops.encode_head(x, extrapolation_mask, self.conv_in.weight, self.conv_in.bias, ld=32, dtype=dt)
        else:
            h = ops.cast(ops.nchw_to_nhwc(x, c_pad=32), dt)
        return self.quant_conv.forward_nhwc(self.encoder.forward_nhwc(h), out_dtype=torch.float32)

    def _decode_nhwc(self, quant_nhwc):
        """fp32 quantised latent (B,h,w,D) -> fp32 RGB-D (B,H,W,4)."""
        q = ops.cast(quant_nhwc, self.compute_dtype)
        return self.decoder.forward_nhwc(self.post_quant_conv.forward_nhwc(q))

    # ---- reference API ----
    def encode(self, x, topk=None, encoding_indices=None, extrapolation_mask=None, use_old=False, sample_number=1):
        pre = self._encode_nhwc(x, extrapolation_mask)
        if not self.use_vq():
            return ops.nhwc_to_nchw(pre)
        if topk is None:
            if encoding_indices is None:
                zq, idx, _ = self.quantize.quantize_nhwc(pre)
            else:
                B, h, w, D = pre.shape
                idx = encoding_indices.reshape(B, h, w)
                zq = ops.vq_gather(self.quantize._codebook()[0], idx).view(B, h, w, D)
            return ops.nhwc_to_nchw(zq), self.quantize.commit_loss_nhwc(pre, idx), (None, None, idx), ops.nhwc_to_nchw(pre)
        zqs, idx = self.quantize.sample_nhwc(pre, topk, sample_number, extrapolation_mask)
        # zqs is (B,S,h,w,D): a sample's slice has batch stride S*h*w*D — the kernels take dense NHWC only
        quants = torch.stack([ops.nhwc_to_nchw(zqs[:, i].contiguous()) for i in range(zqs.shape[1])], 1)
        return quants, None, (None, None, idx), ops.nhwc_to_nchw(pre)

    def decode(self, quant):
        return ops.nhwc_to_nchw(self._decode_nhwc(ops.nchw_to_nhwc(quant)))

    def forward(self, input, topk=None, extrapolation_mask=None, sample_number=1, get_codebook_count=False,
                get_pre_quantized_feature=False, get_quantized_feature=False):
        flags = (bool(get_codebook_count), bool(get_pre_quantized_feature), bool(get_quantized_feature))
        replay_safe = topk is None or (topk == 1 and not self.quantize.consume_host_rng)   # no host RNG in the graph
        if self.use_hip_graph and not getattr(self, "_graph_bypass", False) and replay_safe and input.is_cuda \
                and not torch.is_grad_enabled():
            return self._forward_graphed(input, topk, extrapolation_mask, sample_number, flags)
        return self._forward_guarded(input, topk, extrapolation_mask, sample_number, flags)

    # ---- range guard of the split-fp32 arithmetic (include/sgam_hip.h, sgam_f32x_set_range_flag) ----
    range_check = "sync"     # "sync": eager forwards verify the flag (one device sync) and recompute; "off": caller checks

    def _guard_active(self, input):
        return (self.range_check == "sync" and input.is_cuda and self.compute_dtype == torch.float32
                and ops.F32_MODE == "split" and not torch.cuda.is_current_stream_capturing())

    def _forward_guarded(self, input, topk, extrapolation_mask, sample_number, flags):
        """eager forward; if a split-fp32 kernel reported a non-finite output (an activation beyond fp16's range), the
        forward is recomputed on the fp32-in MFMA path, which has no range precondition, and the process stays there"""
        guard = self._guard_active(input)
        if guard:
            ops.range_flag(input.device)
            rng = torch.get_rng_state() if (topk is not None and topk > 1) else None
        out = self._forward_eager(input, topk, extrapolation_mask, sample_number, *flags)
        if guard and ops.f32x_range_tripped():
            import warnings
            warnings.warn("sgam split-fp32 path: an activation left fp16's range (|x| >= 65520) and produced a non-finite "
                          "value; recomputing on the fp32-in MFMA path and switching SGAM_F32_MODE to 'mfma'", RuntimeWarning)
            ops.set_f32_mode("mfma")
            self._graphs = {}
            if rng is not None:
                torch.set_rng_state(rng)
            out = self._forward_eager(input, topk, extrapolation_mask, sample_number, *flags)
        return out

    def _forward_eager(self, input, topk=None, extrapolation_mask=None, sample_number=1, get_codebook_count=False,
                       get_pre_quantized_feature=False, get_quantized_feature=False):
        pre = self._encode_nhwc(input, extrapolation_mask)
        if not self.use_vq():
            dec = ops.nhwc_to_nchw(self._decode_nhwc(pre))
            return dec, torch.tensor(0).to(dec.device), ops.nhwc_to_nchw(pre)
        want_q = get_quantized_feature
        diff = None
        if topk is None:
            zq, idx, _ = self.quantize.quantize_nhwc(pre)
            diff = self.quantize.commit_loss_nhwc(pre, idx)          # emb_loss (model.py:144-147, quantize.py:296-301)
            decs = ops.nhwc_to_nchw(self._decode_nhwc(zq))
            quants = ops.nhwc_to_nchw(zq) if want_q else None
        else:
            zqs, idx = self.quantize.sample_nhwc(pre, topk, sample_number, extrapolation_mask)
            zs = [zqs[:, i].contiguous() for i in range(sample_number)]   # dense NHWC per sample (batch stride!)
            decs = [ops.nhwc_to_nchw(self._decode_nhwc(z))[None] for z in zs]
            # (B,S,D,h,w) as the channels-first VIEW of the NHWC samples: same shape and values as the reference's stacked
            # tensor, no transpose / stack launches in the frame (the scene loop only files it in the step's result)
            quants = zqs.permute(0, 1, 4, 2, 3) if want_q else None
        res = [decs, diff]
        if get_codebook_count:
            res.append(idx)
        if get_pre_quantized_feature:
            # inference (the scene loop): the channels-first view of the NHWC latent; a training caller gets its own tensor
            res.append(pre.permute(0, 3, 1, 2) if not torch.is_grad_enabled() else ops.nhwc_to_nchw(pre))
        if get_quantized_feature:
            res.append(quants)
        return res

    # ---- training call surface (reference model.py:52-57, 271-345, 405-432) ----
    learning_rate = 4.5e-6          # main.py sets model.learning_rate before fit(); same default as the shipped configs' base rate

    def init_loss(self):
        """`self.loss = instantiate_from_config(lossconfig)` (model.py:57), deferred: the inference path never touches the
        loss (LPIPS trunk + PatchGAN), so it is built on first training use — call this BEFORE load_state_dict when a
        checkpoint's `loss.*` tensors are wanted.  Returns the container (modules/losses/vqperceptual.py)."""
        if getattr(self, "loss", None) is None:
            from ...config import instantiate_from_config
            if not self.lossconfig or not self.lossconfig.get("target"):
                raise ops.SgamHipError("VQModel: lossconfig has no `target` — nothing to train against")
            self.loss = instantiate_from_config(self.lossconfig).to(self.device)
        return self.loss

    def _trainer_for_step(self):
        tr = getattr(self, "_trainer", None)
        if tr is None:
            from ... import training
            loss = self.init_loss()
            if getattr(loss, "use_discriminative_loss", False):
                tr = training.VQGANTrainer(self, loss, phase=self.phase, lr=self.learning_rate)
            else:
                # configure_optimizers returns only opt_ae then (model.py:427-432); the loss reduces to L1 (+ LPIPS) + codebook
                if loss.perceptual_weight > 0:
                    raise NotImplementedError("perceptual loss without the discriminator branch: use training.VQGANTrainer directly")
                tr = training.AutoencoderTrainer(self, phase=self.phase, lr=self.learning_rate, codebook_weight=loss.codebook_weight)
            object.__setattr__(self, "_trainer", tr)        # not a submodule: the trainer refers back to the model
        return tr

    def configure_optimizers(self):
        """model.py:405-432: (opt_ae, opt_disc) — or opt_ae alone without the discriminative loss — as descriptors of the two
        Adam(lr, betas=(0.5, 0.9)) parameter sets; the updates themselves run in csrc/train.hip (`sgam_adam_step_f32`) from
        `training_step`, which — like the reference's manual-optimisation step — owns zero_grad / backward / step."""
        from ... import training
        tr = self._trainer_for_step()
        opt_ae = training.AdamHandle(tr.parameters(), tr.lr, tr.state)
        if isinstance(tr, training.VQGANTrainer):
            return opt_ae, training.AdamHandle(list(tr.disc.parameters()), tr.lr, tr.dstate)
        return opt_ae

    def training_step(self, batch, batch_idx):
        """model.py:271-345 on the HIP backward (sgam_neurips22_amd.training): online k-means refresh, forward, autoencoder
        loss + backward + Adam, discriminator loss + backward + Adam.  Returns aeloss; the logged scalars of the step are
        left in `self.logged` (what the reference hands to `self.log` / `self.log_dict`)."""
        tr = self._trainer_for_step()
        tr.global_step = max(tr.global_step, int(self.global_step))     # (the online k-means refresh runs inside tr.step)
        if self.phase == "conditional_generation":
            x, x_dst, extrapolation_mask, _ = self.get_x(batch, self.data_config["dataset"] if isinstance(self.data_config, dict)
                                                         else self.data_config.dataset, return_extrapolation_mask=True)
        elif self.phase == "codebook":
            x = self.get_input(self.image_key, batch).to(self.device)
            x_dst, extrapolation_mask = x, None
        else:
            raise NotImplementedError(self.phase)
        aeloss, log = tr.step(x, x_dst, extrapolation_mask)
        self.global_step = tr.global_step
        self.logged = dict(log)
        self.logged["train/aeloss"] = aeloss
        return aeloss

    def get_last_layer(self):
        return self.decoder.conv_out.weight

    def get_input(self, key, batch):
        x = batch[key]
        if len(x.shape) == 3:
            x = x[..., None]
        if len(x.shape) == 4:
            x = x.permute(0, 3, 1, 2).to(memory_format=torch.contiguous_format)
        elif len(x.shape) == 5:
            x = x.permute(0, 1, 4, 2, 3).to(memory_format=torch.contiguous_format)
        return x

    def get_x(self, batch, dataset, return_extrapolation_mask=False, no_depth_range=False, parallel=True):
        """Reference model.py:179-269.  Warp (or take the supplied warp), normalise depths, assemble x / x_dst."""
        dev = self.device
        if batch["dst_img"].device != dev:
            for k in batch:
                if hasattr(batch[k], "to"):
                    batch[k] = batch[k].to(device=dev)
        if dataset not in ops.DATASET_NORM:
            raise NotImplementedError
        if 'warped_tgt_features' in batch:
            x_rgb = batch['warped_tgt_features']
            wo = batch.get("_warp_out") if hasattr(batch, "get") else None
            if wo is not None and x_rgb.shape[0] == 1 and x_rgb.data_ptr() == wo["x"].data_ptr():
                # the scene loop warped straight into its persistent model input: the normalised depth and the mask
                # complete it in place (no cat, no copy)
                x = wo["x"]
                warped_depth, extrapolation_mask = ops.depth_normalise(
                    batch['warped_tgt_depth'][:, None], dataset, compute_mask=True, mask_bool=True, out=x[:, 3:4],
                    out_mask=wo["extrap"])
            else:
                wd, extrapolation_mask = ops.depth_normalise(batch['warped_tgt_depth'][:, None], dataset,
                                                             compute_mask=True, mask_bool=True)
                x = torch.cat([x_rgb, wd], 1)
                warped_depth = wd
        else:
            x, extrapolation_mask, warped_depth = splat_to_model_input(
                batch, dataset, depth_range=None if no_depth_range else self.depth_range)
        if "_x_dst" in batch:         # the scene loop's target is a constant (zeros): it assembles x_dst once
            x_dst = batch["_x_dst"]
        else:
            x_dst = self.get_input("dst_img", batch)
            x_depth = self.get_input("dst_depth", batch)
            x_scaled_inverse_depth, _ = ops.depth_normalise(x_depth, dataset, compute_mask=False)
            x_dst = torch.cat([x_dst, x_scaled_inverse_depth], 1)
        if return_extrapolation_mask:
            return x, x_dst, extrapolation_mask, warped_depth
        return x, x_dst
