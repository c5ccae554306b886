"""
hydragen_amd -- MI355X-native (gfx950 / CDNA4) implementation of Hydragen's decode hot path:
decomposed shared-prefix attention (prefix pass on the matrix cores, wavefront-level suffix
pass, log-sum-exp combine) as hand-written HIP kernels behind a C ABI
(include/hydragen_hip.h), exposed with the reference's own Python operator signatures.

    from hydragen_amd.attention import hydragen_attention, hydragen_attention_nopad, combine_lse
    from hydragen_amd.flash import flash_attention, flash_attention_varlen, flash_attention_seqlen
"""

__version__ = "0.1.0"
