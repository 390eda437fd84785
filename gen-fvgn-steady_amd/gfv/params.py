"""Hyper-parameter namespace with the reference's defaults (utils/get_param.py:37-75) for the hot path."""
import argparse


def default_params(**overrides):
    d = dict(net="TransFVGN_v2", n_epochs=210000, batch_size=8, average_sequence_length=500, dataset_size=100,
             lr=5e-5, integrator="imex", norm_uvp=True, norm_global=True, ncn_smooth=True, conserved_form=True,
             max_inner_steps=20, order="2nd", loss_cont=6e4, loss_mom=5e4, loss_press=1.0, hidden_size=128,
             message_passing_num=3, node_phi_size=3, node_input_size=12, node_one_hot=5, node_output_size=3)
    d.update(overrides)
    return argparse.Namespace(**d)
