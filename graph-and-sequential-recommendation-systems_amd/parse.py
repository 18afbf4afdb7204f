"""Flag surface of the reference's parse.py (parse.py:16-114): the same 34 flags
with the same names, types and defaults, plus additive flags (marked NEW) whose
defaults reproduce the reference's behaviour."""
import argparse


def build_parser():
    p = argparse.ArgumentParser(description="Go LightGCN")
    # core training parameters                                    parse.py:20-37
    p.add_argument('--bpr_batch', type=int, default=2048)
    p.add_argument('--recdim', type=int, default=64)
    p.add_argument('--layer', type=int, default=3)
    p.add_argument('--lr', type=float, default=0.001)
    p.add_argument('--decay', type=float, default=1e-4)
    p.add_argument('--dropout', type=int, default=0)
    p.add_argument('--keepprob', type=float, default=0.6)
    p.add_argument('--epochs', type=int, default=1000)
    p.add_argument('--testbatch', type=int, default=100)
    # dataset & paths                                              parse.py:40-45
    p.add_argument('--dataset', type=str, default='gowalla')
    p.add_argument('--checkpoint_dir', type=str, default='./checkpoints')
    p.add_argument('--topks', type=str, default='[20]')
    # logging & reproducibility                                    parse.py:48-61
    p.add_argument('--tensorboard', type=int, default=1)
    p.add_argument('--comment', type=str, default='lgn')
    p.add_argument('--load', type=int, default=0)
    p.add_argument('--pretrain', type=int, default=0)
    p.add_argument('--seed', type=int, default=2020)
    p.add_argument('--model', type=str, default='lgn')
    p.add_argument('--a_fold', type=int, default=100)
    p.add_argument('--A_split', dest='A_split', action='store_true')
    p.add_argument('--no-A_split', dest='A_split', action='store_false')
    p.set_defaults(A_split=False)
    # global smoothing / PPR                                       parse.py:69-74
    p.add_argument('--exp_smooth_beta', type=float, default=0.5)
    p.add_argument('--use_ppr_weights', action='store_true')
    p.add_argument('--ppr_weights_path', type=str, default=None)
    # scheduler                                                    parse.py:77-82
    p.add_argument('--use_scheduler', action='store_true')
    p.add_argument('--sched_milestones', type=str, default='[120,240,360,480]')
    p.add_argument('--sched_gamma', type=float, default=0.5)
    # popularity gate                                              parse.py:85-94
    p.add_argument('--use_pop_gate', action='store_true')
    p.add_argument('--pop_hidden', type=int, default=32)
    p.add_argument('--gate_hidden', type=int, default=64)
    p.add_argument('--gate_entropy_coeff', type=float, default=1e-4)
    p.add_argument('--pop_gate_temp', type=float, default=1.0)
    # item-item adjacency                                          parse.py:97-102
    p.add_argument('--use_item_item', action='store_true')
    p.add_argument('--i2i_path', type=str, default=None)
    p.add_argument('--i2i_alpha', type=float, default=0.0)
    # miscellaneous                                                parse.py:105-112
    p.add_argument('--multicore', type=int, default=0)
    p.add_argument('--resume', action='store_true')
    p.add_argument('--resume_path', type=str, default=None)
    p.add_argument('--save_every', type=int, default=10)
    # ---- NEW (additive) ----------------------------------------------------
    p.add_argument('--sampler', type=str, default='auto', choices=['auto', 'cpp', 'python'],
                   help="NEW: 'cpp' = sampling.cpp stream, 'python' = numpy fallback stream, "
                        "'auto' = cpp unless a user has no positives")
    p.add_argument('--act_dtype', type=str, default='fp32', choices=['fp32', 'bf16', 'fp8'],
                   help='NEW: storage type of propagated layer activations, forward and backward (accumulation, parameters and Adam '
                        'state stay fp32). fp8 = OCP E4M3 with a power-of-two fp32 scale per row; d >= 64, default model only')
    p.add_argument('--xcd_remap', type=int, default=1,
                   help='NEW: give every XCD a contiguous range of graph rows')
    p.add_argument('--row_order', type=str, default='xcd', choices=['natural', 'rcm', 'cocluster', 'xcd'],
                   help='NEW: processing order of graph rows in the SpMM kernels (L2 locality; results unchanged)')
    p.add_argument('--prefetch_epoch', type=int, default=1,
                   help='NEW: 1 (default) = sample/shuffle/upload epoch e+1 while epoch e runs on the GPU (same triplets: the '
                        'sampler and shuffle streams do not depend on training; Gowalla epoch 72.5 -> 61 ms); 0 = at the start '
                        'of each epoch, for code that draws from those RNG streams between epochs')
    p.add_argument('--dense_last', type=str, default='auto', choices=['auto', '0', '1'],
                   help='NEW: last forward layer on the batch rows only (0), densely (1), or whichever is cheaper for '
                        'this graph and batch size (auto)')
    p.add_argument('--fused_variants', type=int, default=1,
                   help='NEW: 1 = --use_pop_gate / --use_item_item train inside the fused HIP step; 0 = autograd path (torch MLPs '
                        'and torch Adam around the HIP propagation kernels)')
    p.add_argument('--hub_nnz', type=int, default=0,
                   help='NEW: graph rows with more non-zeros than this get their last-layer row from a whole-chip SpMM once per '
                        'step instead of from every triplet that names them (0 = library default 131072, < 0 = off)')
    p.add_argument('--gpu_sampler', type=int, default=1,
                   help='NEW: 1 = the cpp-mode BPR sampler runs on the GPU (same rand() stream, same rows); 0 = on the host')
    p.add_argument('--lazy_loss', type=int, default=0,
                   help='NEW: 1 = BPRLoss.stageOne returns a float-like DeferredLoss that is read from the device only when looked at '
                        '(the reference returns loss.cpu().item(): a host round trip per step); 0 = a Python float, as the reference')
    p.add_argument('--gpu_shuffle', type=int, default=1,
                   help='NEW: 1 = the epoch permutation (numpy legacy shuffle: MT19937 + Fisher-Yates, same stream, same permutation) is '
                        'computed on the GPU; 0 = on the host')
    p.add_argument('--eval_fused', type=int, default=1,
                   help='NEW: 1 = Procedure.Test through the fused HIP kernels (MFMA scores + mask + top-k, metrics on device); '
                        '0 = torch matmul/topk harness')
    p.add_argument('--reg_rows', type=str, default='propagated', choices=['propagated', 'ego'],
                   help="NEW: rows the L2 term of bpr_loss is taken on. 'propagated' (default) = this reference (model.py:173: the "
                        "propagated rows of the batch); 'ego' = upstream LightGCN (userEmb0 / posEmb0 / negEmb0, the embedding "
                        "tables' own rows) -- the loss behind the recorded 1000-epoch run and the README table the reference keeps")
    p.add_argument('--data_path', type=str, default=None,
                   help='NEW: directory that holds <dataset>/train.txt (default: <root>/data)')
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)
