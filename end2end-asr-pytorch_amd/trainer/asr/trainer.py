"""Training / validation loop with the reference's Trainer().train(...) signature (reference: trainer/asr/trainer.py).

Differences that keep the semantics: token ids reach the host in ONE copy per step (the reference calls int() on every
element of two (B,T) device tensors, trainer.py:62-75); the loss value is read once; under data parallelism every rank
back-propagates its local loss SUM, the gradients and the [loss sum, token count] statistics are summed by one all-reduce and
the optimiser divides by the global token count (= the reference's loss over the gathered batch, for the update AND for the
logged train loss); validation runs un-sharded on every rank with the local normalisation, so `valid_loss` and the best-model
choice are identical on every rank and to a single-GPU run; the decision to skip a batch with an infinite loss is taken
collectively so that ranks never diverge.
"""
import logging
import os
import sys
import time

import torch
import torch.distributed as dist
from tqdm import tqdm

from utils import constant
from utils.audio import gpu_front_end
from utils.data_loader import DevicePrefetcher
from utils.functions import save_model
from asr_hip import ops
from asr_hip.text import edit_distance_batch
from utils.metrics import calculate_metrics


def _strings(id_rows, id2label):
    out = []
    for row in id_rows:
        s = []
        for x in row:
            if x == constant.PAD_TOKEN:
                break
            s.append(id2label[x])
        out.append("".join(s))
    return out


def _strip(s):
    return s.replace(constant.SOS_CHAR, '').replace(constant.EOS_CHAR, '')


class _Pending:
    """Results of a training step that are still on their way to the host: the token ids (gold, hypothesis) and the loss are
    copied into pinned memory asynchronously; .result() waits for that copy only -- the trainer asks for it AFTER it has
    enqueued the next step, so the strings / CER / WER of step i are computed while the GPU runs step i + 1."""
    _pool = {}

    def __init__(self, gold_seq, hyp_seq, loss, id2label, graph_opt=None):
        self.graph_opt = graph_opt             # the optimiser when the step was a REPLAYED one (its skip is accounted on the host here)
        turn = _Pending._pool["turn"] = (_Pending._pool.get("turn", 0) + 1) & 1        # two buffers per shape: one pending, one filling
        key = (tuple(gold_seq.shape), turn)
        buf = _Pending._pool.get(key)
        if buf is None:
            buf = _Pending._pool[key] = (torch.empty((2,) + tuple(gold_seq.shape), dtype=torch.int64).pin_memory(),
                                         torch.empty(1, dtype=torch.float32).pin_memory(), torch.zeros(1, dtype=torch.int64).pin_memory())
        self.ids, self.loss, self.skip = buf
        self.ids[0].copy_(gold_seq, non_blocking=True)
        self.ids[1].copy_(hyp_seq, non_blocking=True)
        self.loss.copy_(loss.detach().reshape(1).float(), non_blocking=True)
        if graph_opt is not None:
            # the DEVICE's decision travels with the loss (same stream, same event): its guard also cancels a step whose gradient scale
            # is not finite while the loss is (asr_adam_noam_step), which the host could not re-derive from the loss value alone
            from asr_hip import ops
            self.skip.copy_(ops.step_state(loss.device)[2:3], non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self.id2label = id2label

    def result(self):
        self.event.synchronize()
        loss_value = float(self.loss[0])
        bad_loss = loss_value != loss_value or loss_value in (float("inf"), float("-inf"))
        if self.graph_opt is not None and int(self.skip[0]) != 0:
            # the replayed step cancelled itself on the device (non-finite loss or gradient scale): the host mirror of the step count,
            # the rate and Adam's 'step' fields follow the device, whatever the loss value says
            from asr_hip.graph import GraphedTrainStep
            GraphedTrainStep.step_skipped(self.graph_opt)
            if not bad_loss:
                logging.info("The replayed step skipped its update (non-finite gradient scale); weights and moments untouched")
        if bad_loss:
            # reference trainer.py:102-104 skips such a batch; a replayed step cannot branch on the host, so the device-side optimiser
            # step cancelled itself (asr_adam_noam_step guard) and the batch is left out of the running loss here
            logging.info("Found infinity loss, masking (the replayed step left weights and moments untouched)")
            return None
        return (loss_value,) + Trainer._text_metrics(self.ids.tolist(), self.id2label)


class Trainer():
    GRAPH_CACHE = max(1, int(os.environ.get("ASR_GRAPH_CACHE", "12")))     # captured step shapes kept by --graph-buckets (at least one)

    def __init__(self):
        logging.info("Trainer is initialized")

    @staticmethod
    def _frames_after_cnn(T, feat):
        """Encoder positions produced by T input frames (models/asr/transformer.py: vgg_cnn = two 2x2/2 max-pools behind same-padded
        3x3 convolutions; emb_cnn = time kernel 11 / stride 2 / padding 10, then kernel 11 / stride 1 / no padding)."""
        if feat == "vgg_cnn":
            return (int(T) // 2) // 2
        if feat == "emb_cnn":
            return ((int(T) + 20 - 11) // 2 + 1) - 10
        return int(T)

    def _graph_step(self, model, opt, src, src_lengths, tgt, smoothing):
        """--graph-buckets N: the training step as a captured hipGraph per (batch, padded frames) shape.  The batch is copied
        into the graph's static buffers -- time axis zero-padded to a multiple of N frames, targets PAD-padded to --tgt-max-len - 1
        columns (Decoder.preprocess strips PAD, so the targets are unchanged; the encoder positions the extra zero frames add are masked
        through the lengths (clamped to the positions of the batch as collated), so attention sees the un-bucketed batch; what remains
        is the convolutions' view of the last frames of the LONGEST utterance -- followed by zero frames instead of the image border,
        exactly what the collate padding does to every shorter utterance.  With --feat_extractor emb_cnn the BatchNorm batch statistics
        are taken over the batch as collated, collate padding included as in the reference; the frames the bucket adds are masked out
        of them on the device: tests/test_gpu_graph.py) -- and the graph is replayed.  At most
        GRAPH_CACHE shapes stay captured (least recently used first out: a captured step owns its activations' memory pool).  The first batch of a new
        shape runs eagerly (that IS its training step) and captures.  Returns (loss value, gold_seq, hyp_seq) or None when the
        shapes do not fit (falls back to eager launches).  Under --parallel (an active gradient reducer) the step is the four-graph
        form of asr_hip/graph.py with the RCCL all-reduces between the graphs: every rank pads to the same bucket (the sampler
        hands every rank the same bin, so the padded shape is a function of the bin), so all ranks replay the same sequence."""
        from asr_hip.graph import GraphedTrainStep
        a = constant.args
        L = int(a.tgt_max_len) - 1
        if tgt.shape[1] > L or src.dim() != 4:
            return None
        N = int(a.graph_buckets)
        # the front end is the MODEL's (a --continue-from checkpoint brings its own: load_model builds the model from the checkpoint's
        # args, and the command line's --feat_extractor default would clamp an emb_cnn model's lengths to T // 4)
        core = model.module if hasattr(model, "module") else model
        feat = getattr(core, "feat_extractor", getattr(a, "feat_extractor", ""))
        # emb_cnn: the BatchNorm batch statistics run over the batch AS COLLATED (like the reference's, collate padding included); the
        # frames a bucket adds behind it are masked out of them by a device-side length (round 6: asr_bn_batch_stats_v)
        emb_valid = None
        if feat == "emb_cnn":
            t1 = (int(src.shape[3]) + 20 - 11) // 2 + 1
            emb_valid = [t1, t1 - 10]
        B, C, F, T = src.shape
        Tb = (T + N - 1) // N * N
        key = (B, C, F, Tb, L, src.dtype)
        graphs = self.__dict__.setdefault("_graphs", {})
        gs = graphs.pop(key, None)
        if gs is not None:
            graphs[key] = gs                      # most recently used last
        while gs is None and len(graphs) >= self.GRAPH_CACHE:
            old = next(iter(graphs))
            del graphs[old]                       # frees that shape's graph, static buffers and memory pool
            logging.info("graph cache: dropped the captured step of shape %s", (old,))
        lens = torch.as_tensor(src_lengths).to(torch.int32)
        if Tb > T:
            # The model masks encoder position j by j < length with the PRE-CNN lengths (a reference quirk kept on purpose), so on the
            # batch as collated every one of its T' positions below an utterance's length is attended.  The bucket padding adds
            # positions T' .. T'_b - 1 that do not exist in the un-bucketed batch: clamping the lengths to T' masks exactly those
            # (keys and rows), whatever the utterance -- attention sees the batch it would see with --graph-buckets 0.
            lens = torch.clamp(lens, max=self._frames_after_cnn(T, feat))
        if gs is None:
            src_b = torch.zeros((B, C, F, Tb), device=src.device, dtype=src.dtype)
            src_b[..., :T].copy_(src)
            tgt_b = torch.zeros((B, L), device=tgt.device, dtype=tgt.dtype)
            tgt_b[:, :tgt.shape[1]].copy_(tgt)
            gs = graphs[key] = GraphedTrainStep(model, opt, smoothing, src_b, lens, tgt_b,
                                                clip_max_norm=a.max_norm if a.clip else None, warmup_steps=1,
                                                replay_after_capture=False, ddp_graph=getattr(a, "ddp_graph", None), emb_valid=emb_valid)
            if gs.ddp_graph_mode is not None and not self.__dict__.get("_logged_ddp_graph"):
                self._logged_ddp_graph = True
                logging.info("data-parallel graph replay: %s (--ddp-graph %s)", gs.ddp_graph_mode, gs.ddp_graph)
            loss, gold_seq, hyp_seq = gs.warm           # the eager step the constructor ran IS this batch's training step
            return self._global_loss(opt, loss), gold_seq, hyp_seq
        else:
            gs.src[..., :T].copy_(src, non_blocking=True)
            if Tb > T:
                gs.src[..., T:].zero_()
            gs.tgt.zero_()
            gs.tgt[:, :tgt.shape[1]].copy_(tgt, non_blocking=True)
            gs.sync_step_counter()
            gs(src_len=lens, emb_valid=emb_valid)
        return self._global_loss(opt, gs.loss), gs.gold_seq, gs.hyp_seq

    @staticmethod
    def publish_mean_loss(opt, loss):
        """Data-parallel normalisation for a loss that is a MEAN over the local batch (CTC, reference utils/metrics.py:133-154 with
        reduction 'mean'): every rank back-propagates its local mean; the stats slot of the gradient buffer gets [local mean, 1], so
        the one gradient all-reduce also produces [sum of the local means, number of ranks] and the optimiser divides the summed
        gradients by the rank count -- the mean over the gathered batch (the sampler gives every rank the same number of
        utterances), which is what the reference's nn.DataParallel computes.  Call between the loss and backward()."""
        red = getattr(opt.optimizer, "reducer", None)
        if red is None or not red.active:
            return
        st = opt.optimizer.flat.stats
        st[0:1].copy_(loss.detach().reshape(1).float())
        st[1:2].fill_(1.0)

    @staticmethod
    def _global_loss(opt, local_loss):
        """Data parallel: the reference's number is the mean over the gathered batch = all-reduced loss sum / all-reduced token
        count, both in the gradient buffer's stats slot once the step has run (device tensors: no host synchronisation)."""
        red = getattr(opt.optimizer, "reducer", None)
        if red is None or not red.active:
            return local_loss
        st = opt.optimizer.flat.stats
        return st[0] / st[1].clamp_min(1.0)

    def _run_batch(self, model, data, smoothing, loss_type, id2label, opt=None):
        src, tgt, src_percentages, src_lengths, tgt_lengths = data
        if constant.USE_CUDA:
            src, tgt = src.cuda(non_blocking=True), tgt.cuda(non_blocking=True)
        if getattr(constant.args, "gpu_frontend", False):
            a = constant.args
            src, src_lengths = gpu_front_end(src, src_lengths, a.sample_rate, a.window_size, a.window_stride, a.src_max_len)
        if opt is not None and loss_type == "ce" and getattr(constant.args, "graph_buckets", 0) > 0 and src.is_cuda:
            r = self._graph_step(model, opt, src, src_lengths, tgt, smoothing)
            if r is not None:
                loss, gold_seq, hyp_seq = r
                return _Pending(gold_seq, hyp_seq, loss, id2label, graph_opt=opt)         # asynchronous D2H; no sync in this step
        if opt is not None:
            opt.zero_grad()
        pred, gold, hyp_seq, gold_seq = model(src, src_lengths, tgt, verbose=False)
        if loss_type == "ctc":
            # reference trainer.py:81-85: input lengths = source percentages x decoder positions, targets' true lengths
            sizes = (src_percentages.float() * int(pred.size(1))).int()
            loss, sums = calculate_metrics(pred, gold, input_lengths=sizes, target_lengths=tgt_lengths, loss_type="ctc")
            if opt is not None:
                self.publish_mean_loss(opt, loss)
        else:
            loss, sums = calculate_metrics(pred, gold, smoothing=smoothing, loss_type=loss_type, sync=False)
        finite = torch.isfinite(loss.detach()).float()
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(finite, op=dist.ReduceOp.MIN)        # skip on every rank or on none
        if finite.item() == 0:
            logging.info("Found infinity loss, masking")
            return None
        loss_value = None
        if opt is not None:
            ops.backward_from(loss)
            if constant.args.clip:
                opt.optimizer.clip_grad_norm_(constant.args.max_norm)
            opt.step()
            red = getattr(opt.optimizer, "reducer", None)
            if red is not None and red.active:
                # data parallel: `loss` is this rank's local mean; the reference's number is the mean over the gathered
                # batch = all-reduced loss sum / all-reduced token count (the stats slot of the gradient buffer)
                loss_value = opt.optimizer.global_loss()
        if opt is not None and loss_value is None and gold_seq.is_cuda:
            return _Pending(gold_seq, hyp_seq, loss, id2label)  # training: the ids / loss reach the host under the next step
        ids = torch.stack([gold_seq, hyp_seq]).cpu().tolist()   # one D2H copy
        return ((loss.item() if loss_value is None else loss_value),) + self._text_metrics(ids, id2label)

    @staticmethod
    def _text_metrics(ids, id2label):
        """(gold ids, hypothesis ids) of a batch -> (CER distance, WER distance, #characters, #words) (reference trainer.py:62-75)"""
        strs_gold, strs_hyps = _strings(ids[0], id2label), _strings(ids[1], id2label)
        # CER / WER of the whole batch in two calls of the native Levenshtein (asr_hip/text.py; per utterance it is
        # calculate_cer(h without spaces, g without spaces) and calculate_wer(h, g) of utils/metrics.py)
        golds, hyps = [_strip(g) for g in strs_gold], [_strip(h) for h in strs_hyps]
        cer = sum(edit_distance_batch([(h.replace(' ', ''), g.replace(' ', '')) for g, h in zip(golds, hyps)]))
        wer = sum(edit_distance_batch([(h.split(), g.split()) for g, h in zip(golds, hyps)]))
        chars = sum(len(g.replace(' ', '')) for g in golds)
        words = sum(len(g.split(" ")) for g in golds)
        return cer, wer, chars, words

    def train(self, model, train_loader, train_sampler, valid_loader_list, opt, loss_type, start_epoch, num_epochs, label2id,
              id2label, last_metrics=None):
        history = []
        best_valid_loss = 1000000000 if last_metrics is None else last_metrics['valid_loss']
        smoothing = constant.args.label_smoothing
        logging.info("name " + constant.args.name)
        rank0 = not dist.is_initialized() or dist.get_rank() == 0
        for epoch in range(start_epoch, num_epochs):
            sys.stdout.flush()
            total_loss = total_cer = total_wer = total_char = total_word = 0
            n_batches = frames = 0
            t0 = time.time()
            logging.info("TRAIN")
            model.train()
            # batches arrive on the device one step ahead (pinned staging + copy stream, utils/data_loader.py)
            feed = DevicePrefetcher(train_loader, torch.device("cuda", torch.cuda.current_device()) if constant.USE_CUDA else None)
            pbar = tqdm(iter(feed), leave=True, total=len(train_loader), disable=not rank0)
            def account(r, i):
                nonlocal total_loss, total_cer, total_wer, total_char, total_word, n_batches
                if r is None:                    # a masked (non-finite) batch
                    return
                loss, cer, wer, chars, words = r
                total_loss += loss; total_cer += cer; total_wer += wer; total_char += chars; total_word += words
                n_batches += 1
                pbar.set_description("(Epoch {}) TRAIN LOSS:{:.4f} CER:{:.2f}% LR:{:.7f} {:.0f} frames/s".format(
                    epoch + 1, total_loss / (i + 1), total_cer * 100 / max(1, total_char), opt._rate, frames / (time.time() - t0)))

            pending = None                       # (--graph-buckets) the previous step's results, still in flight to the host
            for i, data in enumerate(pbar):
                r = self._run_batch(model, data, smoothing, loss_type, id2label, opt)
                if pending is not None:
                    account(pending[0].result(), pending[1])
                    pending = None
                if r is None:
                    continue
                frames += int(data[3].sum())
                if isinstance(r, _Pending):
                    pending = (r, i)
                else:
                    account(r, i)
            if pending is not None:
                account(pending[0].result(), pending[1])
            logging.info("(Epoch {}) TRAIN LOSS:{:.4f} CER:{:.2f}% LR:{:.7f}".format(
                epoch + 1, total_loss / max(1, len(train_loader)), total_cer * 100 / max(1, total_char), opt._rate))

            logging.info("VALID")
            model.eval()
            total_valid_loss = total_valid_cer = total_valid_wer = 0
            n_valid = 1
            for ind, valid_loader in enumerate(valid_loader_list):
                total_valid_loss = total_valid_cer = total_valid_wer = total_valid_char = total_valid_word = 0
                n_valid = max(1, len(valid_loader))
                vbar = tqdm(iter(valid_loader), leave=True, total=len(valid_loader), disable=not rank0)
                for i, data in enumerate(vbar):
                    with torch.no_grad():
                        r = self._run_batch(model, data, smoothing, loss_type, id2label, None)
                    if r is None:
                        continue
                    loss, cer, wer, chars, words = r
                    total_valid_loss += loss; total_valid_cer += cer; total_valid_wer += wer
                    total_valid_char += chars; total_valid_word += words
                    vbar.set_description("VALID SET {} LOSS:{:.4f} CER:{:.2f}%".format(
                        ind, total_valid_loss / (i + 1), total_valid_cer * 100 / max(1, total_valid_char)))
                logging.info("VALID SET {} LOSS:{:.4f} CER:{:.2f}%".format(
                    ind, total_valid_loss / n_valid, total_valid_cer * 100 / max(1, total_valid_char)))

            metrics = {"train_loss": total_loss / max(1, len(train_loader)), "valid_loss": total_valid_loss / n_valid,
                       "train_cer": total_cer, "train_wer": total_wer, "valid_cer": total_valid_cer,
                       "valid_wer": total_valid_wer, "history": history}
            history.append(metrics)
            if epoch % constant.args.save_every == 0:
                save_model(model, epoch + 1, opt, metrics, label2id, id2label, best_model=False)
            if best_valid_loss > total_valid_loss / n_valid:
                best_valid_loss = total_valid_loss / n_valid
                save_model(model, epoch + 1, opt, metrics, label2id, id2label, best_model=True)
            if constant.args.shuffle:
                logging.info("SHUFFLE")
                train_sampler.shuffle(epoch)
