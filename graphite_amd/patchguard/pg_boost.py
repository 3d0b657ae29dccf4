"""Drop-in for the reference's `patchguard/pg_boost.py:13-217` (same signature; 5-tuple for model_type 'CIFAR', :215-216).

Differences from boost.py that this twin keeps: the BASELINE pg parameter table (:45), normalised network input, add_noise
with clipping for the CIFAR family (:201-204), no best_theta in the returned tuple.  The reference also pickles its own
arguments -- the model included -- into ./dumped_data.pkl on every call (:29-43, a debugging leftover); that side effect is
not reproduced.
"""
from .. import boost as _B


def boost(model, x0_large, y0, mask, target_example, target=None, beta=1, iterations=30, pt_file='images/Points/14.csv',
          num_xforms=1000, net_size=32, model_type='CIFAR', goal=None, eta=500, bt=False, budget_factor=200,
          init_theta=None, square_mask=False, early_boost_exit=False, _evaluator=None):
    assert model_type == 'CIFAR'
    adv, tr, perturb, queries, _best_theta, adv_small = _B.boost(
        model, x0_large, y0, mask, target_example, target, beta, iterations, pt_file, num_xforms, net_size, 'PATCHGUARD',
        goal, eta, bt, budget_factor, init_theta, square_mask, early_boost_exit, _evaluator=_evaluator)
    return adv, tr, perturb, queries, adv_small
