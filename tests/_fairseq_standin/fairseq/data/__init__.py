import torch.utils.data


class FairseqDataset(torch.utils.data.Dataset):
    def num_tokens(self, index):
        raise NotImplementedError

    def size(self, index):
        raise NotImplementedError

    def ordered_indices(self):
        raise NotImplementedError

    def collater(self, samples):
        raise NotImplementedError
