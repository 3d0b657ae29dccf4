"""Command-line flags -- same flags, defaults and derived values as the reference's
`parsearguments.py:4-76`.  The reference re-parses `sys.argv` from inside library functions
(generate_mask.py:35,98); `generate_mask` here does the same through this module, but tolerates
foreign arguments (parse_known_args) so that it can be called from any program.
"""
import argparse


def build_parser():
    p = argparse.ArgumentParser(description='GRAPHITE')
    p.add_argument('--victim_id', '-v', default='14')
    p.add_argument('--target_id', '-t', default='1')
    p.add_argument('--tr_lo', type=float, default=0.25)
    p.add_argument('--tr_hi', type=float, default=0.5)
    p.add_argument('--scorefile', '-s', default='score.py')
    p.add_argument('--network', '-n', default='GTSRB')
    p.add_argument('--heatmap', default='Target')
    p.add_argument('--boost_transforms', '-b', type=int, default=100)
    p.add_argument('--mask_transforms', '-m', type=int, default=100)
    p.add_argument('--coarse_mode', default='binary')
    p.add_argument('--heatmap_file')
    p.add_argument('--num_test_xforms', type=int, default=-1)
    p.add_argument('--max_mask_size', type=int, default=-1)
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--joint_iters', type=int, default=1)
    p.add_argument('--image_id', default='')
    p.add_argument('--img_v')
    p.add_argument('--img_t')
    p.add_argument('--hull')
    p.add_argument('--pt_file')
    p.add_argument('--out_path', default='outputs/')
    p.add_argument('--vic_license_plate')
    p.add_argument('--tar_license_plate')
    p.add_argument('--border_outer')
    p.add_argument('--border_inner')
    p.add_argument('--tag', default='')
    p.add_argument('--bt', action='store_true')
    p.add_argument('--square_x')
    p.add_argument('--square_y')
    p.add_argument('--square_size')
    p.add_argument('--early_boost_exit', action='store_true')
    return p


def getarguments(argv=None, strict=False):
    p = build_parser()
    args = p.parse_args(argv) if strict else p.parse_known_args(argv)[0]
    args.coarse_error = 1 - args.tr_hi
    args.reduce_error = 1 - args.tr_lo
    assert args.coarse_error <= args.reduce_error
    assert args.coarse_error >= 0
    assert args.reduce_error <= 1
    assert args.heatmap in ['Target', 'Victim', 'Random']
    base = 'inputs/' + args.network + "/"
    if args.img_v is None:
        args.img_v = base + 'images/' + args.victim_id + ".png"
    if args.img_t is None:
        args.img_t = base + 'images/' + args.target_id + ".png"
    args.mask = args.hull if args.hull is not None else base + 'Hulls/' + args.victim_id + ".png"
    args.lbl_v = int(args.victim_id)
    args.lbl_t = int(args.target_id)
    if args.pt_file is None:
        try:
            pt_file = base + 'Points/' + args.victim_id + ".csv"
            open(pt_file).close()
            args.pt_file = pt_file
        except OSError:
            args.pt_file = None
    return args
