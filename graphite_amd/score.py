"""Default mask-scoring plug-in -- same contract as the reference's `score.py:8-12`.

`generate_mask` imports the scoring module BY FILE NAME (generate_mask.py:32-34), so a user's own
`score.py` keeps working; this is the built-in default.  Note the third argument is the ERROR rate
(1 - transform robustness).
"""


def score_fn(theta, mask, tr_err, object_size, coord=None, threshold=0.75, lbd=5, *args, **kargs):
    if tr_err > threshold:
        return 10000000
    return lbd * mask.sum() / 3 / (object_size) + tr_err
