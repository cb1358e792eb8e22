"""Evaluation entry point -- same command line as the reference (reference: test.py): loads --continue-from, decodes
every test utterance (greedy, or beam search with --beam-search) and reports CER / WER."""
import torch
from tqdm import tqdm

from utils import constant
from utils.metrics import calculate_cer, calculate_cer_en_zh, calculate_wer


def evaluate(model, test_loader, lm=None):
    """reference: test.py:19-62"""
    args = constant.args
    model.eval()
    total_word = total_char = total_cer = total_wer = 0
    total_en_cer = total_zh_cer = total_en_char = total_zh_char = 0
    with torch.no_grad():
        pbar = tqdm(iter(test_loader), leave=True, total=len(test_loader))
        for data in pbar:
            src, tgt, _, src_lengths, _ = data
            if constant.USE_CUDA:
                src, tgt = src.cuda(), tgt.cuda()
            if getattr(args, "gpu_frontend", False):
                from utils.audio import gpu_front_end
                src, src_lengths = gpu_front_end(src, src_lengths, args.sample_rate, args.window_size, args.window_stride,
                                                 args.src_max_len)
            _, strs_hyps, strs_gold = model.evaluate(src, src_lengths, tgt, beam_search=args.beam_search,
                                                     beam_width=args.beam_width, beam_nbest=args.beam_nbest, lm=lm,
                                                     lm_rescoring=args.lm_rescoring, lm_weight=args.lm_weight,
                                                     c_weight=args.c_weight, verbose=args.verbose)
            for hyp, gold in zip(strs_hyps, strs_gold):
                for ch in (constant.EOS_CHAR, constant.SOS_CHAR, constant.PAD_CHAR):
                    hyp, gold = hyp.replace(ch, ""), gold.replace(ch, "")
                total_wer += calculate_wer(hyp, gold)
                total_cer += calculate_cer(hyp.strip(), gold.strip())
                en_cer, zh_cer, n_en, n_zh = calculate_cer_en_zh(hyp, gold)
                total_en_cer += en_cer; total_zh_cer += zh_cer; total_en_char += n_en; total_zh_char += n_zh
                total_word += len(gold.split(" "))
                total_char += len(gold)
            pbar.set_description("TEST CER:{:.2f}% WER:{:.2f}% CER_EN:{:.2f}% CER_ZH:{:.2f}%".format(
                total_cer * 100 / max(1, total_char), total_wer * 100 / max(1, total_word),
                total_en_cer * 100 / max(1, total_en_char), total_zh_cer * 100 / max(1, total_zh_char)))
    return total_cer / max(1, total_char), total_wer / max(1, total_word)


if __name__ == '__main__':
    from utils.data_loader import AudioDataLoader, BucketingSampler, SpectrogramDataset
    from utils.functions import load_model
    args = constant.args
    if args.lm_rescoring:
        raise SystemExit("LM rescoring is outside the accelerated path (SURVEY.md section 2, rows 12-14)")
    model, opt, epoch, metrics, loaded_args, label2id, id2label = load_model(args.continue_from)
    if getattr(loaded_args, "parallel", False):
        print("unwrap data parallel")
        model = model.module
    constant.args.tgt_max_len = max(constant.args.tgt_max_len, 301)      # greedy/beam search always run 300 steps
    audio_conf = dict(sample_rate=loaded_args.sample_rate, window_size=loaded_args.window_size,
                      window_stride=loaded_args.window_stride, window=loaded_args.window, noise_dir=loaded_args.noise_dir,
                      noise_prob=loaded_args.noise_prob, noise_levels=(loaded_args.noise_min, loaded_args.noise_max))
    test_data = SpectrogramDataset(audio_conf=audio_conf, manifest_filepath_list=args.test_manifest_list, label2id=label2id,
                                   normalize=True, augment=False)
    test_sampler = BucketingSampler(test_data, batch_size=args.batch_size)
    test_loader = AudioDataLoader(test_data, num_workers=args.num_workers, batch_sampler=test_sampler)
    print(model)
    evaluate(model, test_loader)
