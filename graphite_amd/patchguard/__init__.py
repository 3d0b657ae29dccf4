"""The PatchGuard flavour of the TR scoring path -- drop-in for the reference's `patchguard/` directory.

  pg_transforms      patchguard/pg_transforms.py      CIFAR-sized frames, always normalised to [-0.5, 0.5], own parameter table
  pg_generate_mask   patchguard/pg_generate_mask.py   mask generation on the baseline pg table
  pg_boost           patchguard/pg_boost.py           RGF boosting on the baseline pg table (5-tuple for CIFAR)
  main_patchguard    patchguard/main_patchguard.py    attack driver
  patchguard_predict patchguard/patchguard_predict.py the robust classifier: bicubic x6 + Normalize (reference code) as a HIP kernel,
                                                      BagNet-17 and masking_defense (third-party, restated from the
                                                      publications, parity unpinned) on the MFMA conv kernels / a HIP epilogue

All pixel work and the batched evaluation are the same HIP path as the GTSRB / CIFAR flavours (model_type 'PATCHGUARD'
inside graphite_amd selects normalisation and the pg parameter table).
"""
