"""Command-line options: the reference's flag names, types and defaults (opts.py:3-269), defined
from a table so that bash_scripts/run_joint.sh / run_att.sh / run_fc_con.sh drive this
implementation unchanged.  Flags the reference parses but never reads (optim*, gate_type,
closest_*, soft_cider) are accepted and ignored the same way."""
import argparse

S, I, F = str, int, float

# (flag, type, default)
_FLAGS = [
    # data
    ('input_json', S, 'data/coco.json'), ('input_fc_dir', S, 'data/cocotalk_fc'),
    ('input_att_dir', S, 'data/cocotalk_att'), ('input_label_h5', S, 'data/coco_label.h5'),
    ('start_from', S, None), ('initialize_retrieval', S, None), ('cached_tokens', S, 'corpus'),
    ('cider_optimization', F, 0),
    # speaker
    ('caption_model', S, 'show_tell'), ('rnn_size', I, 512), ('num_layers', I, 1), ('rnn_type', S, 'lstm'),
    ('input_encoding_size', I, 512), ('att_hid_size', I, 512), ('fc_feat_size', I, 2048),
    ('att_feat_size', I, 2048), ('use_bn', I, 0), ('decoding_constraint', I, 0),
    # optimisation
    ('max_epochs', I, -1), ('batch_size', I, 16), ('grad_clip', F, 0.1), ('drop_prob_lm', F, 0.5),
    ('seq_per_img', I, 1), ('beam_size', I, 1),
    # alternating / joint training
    ('is_alternating', I, 0), ('use_gen_cider_scores', I, 0),
    ('speaker_stage_2_optimizer_path', S, 'optimizer.pth'), ('speaker_stage_2_model_path', S, 'model.pth'),
    ('listener_stage_1_model_path', S, 'model.pth'),
    ('gumbel_temp', F, 10.0), ('multinomial_temp', F, 1.0), ('phase', F, None),
    ('prob_gumbel_softmax', F, 0.25), ('prob_multinomial_soft', F, 0.25),
    ('gumbel_temperature_annealing_factor', F, 0), ('gumbel_temperature_annealing_rate', I, 15),
    ('num_iteration_for_annealing', I, 500),
    # optimiser / schedules
    ('optim', S, 'adam'), ('learning_rate', F, 4e-4), ('learning_rate_decay_start', I, -1),
    ('learning_rate_decay_every', I, 3), ('learning_rate_decay_rate', F, 0.8), ('optim_alpha', F, 0.9),
    ('optim_beta', F, 0.999), ('optim_epsilon', F, 1e-8), ('weight_decay', F, 0),
    ('softmax_cooling_decay_factor', F, 0), ('scheduled_sampling_start', I, -1),
    ('scheduled_sampling_increase_every', I, 5), ('scheduled_sampling_increase_prob', F, 0.05),
    ('scheduled_sampling_max_prob', F, 0.25), ('retrieval_reward_weight_decay_start', I, -1),
    ('retrieval_reward_weight_decay_every', I, 15), ('retrieval_reward_weight_decay_rate', F, 0.8),
    ('gate_type', S, 'softmax'), ('closest_num', I, 10), ('closest_file', S, 'data/closest.pkl'),
    # evaluation / checkpoints
    ('val_images_use', I, 3200), ('save_checkpoint_every', I, 2500), ('checkpoint_path', S, 'save'),
    ('language_eval', I, 0), ('rank_eval', I, 0), ('losses_log_every', I, 1000), ('load_best_score', I, 1),
    ('id', S, ''), ('train_only', I, 0), ('start_with_checkpoint', I, 0),
    # listener
    ('vse_model', S, 'None'), ('vse_rnn_type', S, 'gru'), ('vse_margin', F, 0.2), ('vse_embed_size', I, 1024),
    ('vse_num_layers', I, 1), ('vse_max_violation', I, 1), ('vse_measure', S, 'cosine'), ('vse_use_abs', I, 0),
    ('vse_no_imgnorm', I, 0), ('vse_loss_type', S, 'contrastive'), ('vse_pool_type', S, 'last'),
    # discriminative reward
    ('retrieval_reward', S, 'gumbel'), ('retrieval_reward_weight', F, 0), ('only_one_retrieval', S, 'off'),
    ('share_embed', I, 0), ('caption_loss_weight', F, 1), ('vse_loss_weight', F, 0),
    ('vse_eval_criterion', S, 'rsum'), ('reinforce_baseline_type', S, 'greedy'),
    ('soft_cider', I, 0), ('df', S, 'coco-val'), ('dataset', S, 'coco'),
]


def build_parser():
    p = argparse.ArgumentParser(description='cooperative image captioning on MI355X (reference-compatible flags)')
    for name, typ, default in _FLAGS:
        p.add_argument('--' + name, type=typ, default=default)
    p.add_argument('--alternating_turn', action='append')                       # opts.py:82-83
    p.add_argument('--rank_on_gen_captions', action='store_true')               # opts.py:86-87
    p.add_argument('--continue_from_existing_models', action='store_false')     # opts.py:88-89 (store_false!)
    # additions of this implementation (absent from the reference)
    p.add_argument('--compute_dtype', type=str, default='f32', choices=['f32', 'bf16'],
                   help="speaker arithmetic: f32 = the reference's; bf16 = bf16 operands in the batched products and bf16 "
                        "storage of the region features (f32 accumulation, f32 softmax / losses / optimiser)")
    p.add_argument('--synthetic', type=int, default=0, help='1: COCO-shaped synthetic batches (no dataset needed)')
    p.add_argument('--synthetic_pool', type=int, default=8,
                   help='distinct synthetic batches served round robin (0: draw a fresh batch on every call)')
    p.add_argument('--max_iterations', type=int, default=-1, help='stop after this many iterations (-1: never)')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--eval_at_checkpoint', type=int, default=1,
                   help='1 (the reference): evaluate the val split at every checkpoint and keep model-best / model_vse-best; '
                        '0: save only')
    p.add_argument('--loader_seed', type=int, default=0, help='seed of the per-epoch shuffle shared by the ranks of a '
                                                               'data-parallel run (dataloader.DataLoader)')
    p.add_argument('--prefetch', type=int, default=1, help='1: upload the next batch on a copy stream while the '
                                                            'current step computes (prefetch.PrefetchLoader)')
    return p


def parse_opt(argv=None):
    args = build_parser().parse_args(argv)
    # the reference's sanity checks (opts.py:256-267)
    assert args.rnn_size > 0 and args.num_layers > 0 and args.input_encoding_size > 0 and args.batch_size > 0
    assert 0 <= args.drop_prob_lm < 1, 'drop_prob_lm should be between 0 and 1'
    assert args.seq_per_img > 0 and args.beam_size > 0
    assert args.save_checkpoint_every > 0 and args.losses_log_every > 0
    assert args.language_eval in (0, 1) and args.load_best_score in (0, 1) and args.train_only in (0, 1)
    return args
