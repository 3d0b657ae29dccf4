"""RGF boosting -- drop-in for the reference's `boost.py:9-195` (same signature, same 6-tuple).

Per iteration the reference evaluates q = 10 random directions one after the other (:87-101); the
directions do not depend on each other's results, so they are drawn up front (same torch RNG
stream, same order) and evaluated in ONE fused launch (10 candidates x T transforms); without
back-tracking the step's own evaluation and the NEXT iteration's directions share a launch (11 candidates).
Everything else -- gradient accumulation order, back-tracking ladder, budget rule, query accounting --
follows the reference step by step.
"""
import torch

from .evaluator import TREvaluator, make_evaluator
from .transforms import draw_transforms, normalizes


def boost(model, x0_large, y0, mask, target_example, target=None, beta=1, iterations=30, pt_file='images/Points/14.csv',
          num_xforms=1000, net_size=32, model_type='GTSRB', goal=None, eta=500, bt=False, budget_factor=200,
          init_theta=None, square_mask=False, early_boost_exit=False, _evaluator=None):
    xforms = draw_transforms(num_xforms, model_type)       # 'PATCHGUARD': the baseline pg table (pg_boost.py:45)
    ev = make_evaluator(_evaluator, model, x0_large, xforms, pt_file, model_type, y0, target)
    query_count = 0
    x0 = ev.resize(x0_large, mask.size()[1], mask.size()[2])
    theta = init_theta * mask if init_theta is not None else (target_example - x0) * mask

    success_rate, query_ct = ev.error(mask, theta)
    query_count += query_ct
    best_theta, best_eps = theta, success_rate
    print("==========> Found success rate %.4f using %d queries" % (best_eps, query_count))

    theta, eps = best_theta.clone(), best_eps
    opt_count = 0
    iters_taken = 0
    q = 10
    # Without back-tracking the step of iteration i is taken whatever its evaluation says (:151-160), so the q directions
    # of iteration i+1 (drawn from the same RNG stream right after it) can ride in the SAME launch as the step's own
    # evaluation: 11 candidates instead of two dependent launches of 1 and 10.  Only when the loop is known to go on
    # (the budget rule counts queries, which are deterministic) and the evaluator may speculate.
    fuse = (not bt) and getattr(ev, "speculate", False)
    carried = None                                           # (us, ttts, results) evaluated ahead by the previous launch

    def draw(th):
        us, ttts = [], []
        for _ in range(q):                                   # same RNG draws, same order as :88-90
            u = torch.randn(th.size()).type(torch.FloatTensor) * mask
            u = u / torch.norm(u)
            us.append(u)
            ttts.append(th + beta * u)
        return us, ttts

    for i in range(iterations):
        iters_taken += 1
        gradient = torch.zeros(theta.size())
        min_g1 = float('inf')
        min_ttt = None
        if carried is None:
            us, ttts = draw(theta)
            res = ev.errors([mask] * q, ttts)                # one launch for the q directions
        else:
            us, ttts, res = carried
            carried = None
        for u, ttt, (eps_ttt, opt_ct) in zip(us, ttts, res):
            opt_count += opt_ct
            gradient += (eps_ttt - eps) / beta * u
            if bt and eps_ttt < min_g1:
                min_g1, min_ttt = eps_ttt, ttt
        gradient = 1.0 / q * gradient

        if bt:
            min_theta, min_new_eps = theta, eps
            alpha = 500
            for _ in range(5):
                new_theta = theta - alpha * gradient
                new_eps, opt_ct = ev.error(mask, new_theta)
                opt_count += opt_ct
                alpha = alpha * 2
                if new_eps < min_new_eps:
                    min_theta, min_new_eps = new_theta, new_eps
                else:
                    break
            if min_new_eps >= eps:
                for _ in range(5):
                    alpha = alpha * 0.25
                    new_theta = theta - alpha * gradient
                    new_eps, opt_ct = ev.error(mask, new_theta)
                    opt_count += opt_ct
                    if new_eps < eps:
                        min_theta, min_new_eps = new_theta, new_eps
                        break
            if min_new_eps == eps:
                min_theta = theta - eta * gradient
                min_new_eps, opt_ct = ev.error(mask, min_theta)
                opt_count += opt_ct
            if min_new_eps <= min_g1:
                new_theta, new_eps = min_theta, min_new_eps
            else:
                new_theta, new_eps = min_ttt, min_g1
        else:
            new_theta = theta - eta * gradient
            # will the loop go on after this step?  (:163 with this step's num_xforms queries added; :165 depends on the
            # step's own result, so that exit is never speculated over)
            goes_on = (i + 1 < iterations and
                       not (opt_count + num_xforms + query_count + num_xforms * 11) > (budget_factor * num_xforms) and
                       not (square_mask and i == 0 and early_boost_exit))
            if fuse and goes_on:
                nus, nttts = draw(new_theta)
                both = ev.errors([mask] * (q + 1), [new_theta] + nttts)
                (new_eps, opt_ct), carried = both[0], (nus, nttts, both[1:])
            else:
                new_eps, opt_ct = ev.error(mask, new_theta)
            opt_count += opt_ct

        if new_eps < best_eps:
            best_theta, best_eps = new_theta.clone(), new_eps
        theta, eps = new_theta.clone(), new_eps

        if (opt_count + query_count + num_xforms * 11) > (budget_factor * num_xforms):
            break
        if square_mask and i == 0 and best_eps >= 0.95 and early_boost_exit:
            break
    assert carried is None, "a speculated direction batch was left unused"

    # boost.py:180-183 clips for GTSRB only; pg_boost.py:201-204 clips for its (CIFAR-sized, normalised) family
    adv_example = ev.add_noise(x0_large, mask, 1.0, best_theta, clip=normalizes(model_type))
    mask_large = ev.resize(mask, x0_large.size()[1], x0_large.size()[2])
    perturb = torch.where(mask_large > 0.0, adv_example, torch.ones(adv_example.size()))
    print("final iters_taken: ", iters_taken)
    return adv_example, (1 - best_eps), perturb, query_count + opt_count, best_theta, torch.clamp(x0 + best_theta, 0, 1)
