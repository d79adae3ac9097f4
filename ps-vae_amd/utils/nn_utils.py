"""Config builders with the reference's names and dict shapes
(src/utils/nn_utils.py:139-206)."""

PROPS = ['qed', 'sa', 'logp']   # src/evaluation/utils.py:122


def map_prop_to_idx(props):
    """src/evaluation/utils.py:123-134."""
    p2i = {p: i for i, p in enumerate(PROPS)}
    idxs = []
    for p in props:
        if p not in p2i:
            raise ValueError('Invalid property')
        idxs.append(p2i[p])
    return sorted(list(set(idxs)))


def common_config(args):
    return {'lr': args.lr, 'alpha': args.alpha, 'beta': args.beta, 'max_beta': args.max_beta,
            'step_beta': args.step_beta, 'kl_warmup': args.kl_warmup, 'kl_anneal_iter': args.kl_anneal_iter,
            'selected_properties': map_prop_to_idx(args.props)}


def predictor_config(args):
    return {'predictor': {'dim_feature': args.latent_dim, 'dim_hidden': args.predictor_hidden_dim,
                          'num_property': len(args.props)}}


def encoder_config(args, vocab):
    dim_in = vocab.num_atom_type()
    if hasattr(vocab, 'num_formal_charge'):
        dim_in += vocab.num_formal_charge()
    return {'encoder': {'dim_in': dim_in, 'num_edge_type': vocab.num_bond_type(),
                        'dim_hidden': args.node_hidden_dim, 'dim_out': args.graph_embedding_dim}}


def ps_vae_config(args, tokenizer):
    chem_vocab = tokenizer.chem_vocab
    atom_dim = args.atom_embedding_dim + args.piece_embedding_dim + args.pos_embedding_dim
    return {
        'encoder': {'dim_in': atom_dim, 'num_edge_type': chem_vocab.num_bond_type(),
                    'dim_hidden': args.node_hidden_dim, 'dim_out': args.graph_embedding_dim},
        'vae_piece_decoder': {
            'atom_embedding_dim': args.atom_embedding_dim, 'piece_embedding_dim': args.piece_embedding_dim,
            'max_pos': args.max_pos, 'pos_embedding_dim': args.pos_embedding_dim,
            'piece_hidden_dim': args.piece_hidden_dim, 'node_hidden_dim': args.node_hidden_dim,
            'num_edge_type': chem_vocab.num_bond_type(), 'cond_dim': args.graph_embedding_dim,
            'latent_dim': args.latent_dim, 'tokenizer': tokenizer, 't': 4},
    }


def default_args(**overrides):
    """argparse defaults of src/train.py:57-111 with the hyper-parameters of scripts/train.sh:14-36."""
    import types
    d = dict(lr=1e-3, alpha=0.1, beta=0.0, max_beta=0.01, step_beta=0.002, kl_warmup=0, kl_anneal_iter=1000,
             props=['qed', 'logp'], latent_dim=56, predictor_hidden_dim=200, node_hidden_dim=300,
             graph_embedding_dim=400, atom_embedding_dim=50, piece_embedding_dim=100, pos_embedding_dim=50,
             max_pos=50, piece_hidden_dim=200, batch_size=32, grad_clip=10.0, epochs=6, patience=3)
    d.update(overrides)
    return types.SimpleNamespace(**d)


def build_config(args, tokenizer):
    """The dict assembled at src/train.py:130-132 (ps_vae_config overwrites 'encoder')."""
    config = {**common_config(args), **encoder_config(args, tokenizer.chem_vocab), **predictor_config(args)}
    config.update(ps_vae_config(args, tokenizer))
    return config
